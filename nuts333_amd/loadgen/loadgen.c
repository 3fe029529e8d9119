/*
 * loadgen.c -- closed-loop synthetic telnet-client load generator for a NUTS 3.3.3
 * talker (the reference at /root/reference, or our restatement oracle/talker_port).
 *
 * Why it looks the way it does (each rule is a reference behaviour):
 *   - one input line per TCP segment and never more than one un-acknowledged line per
 *     sender: the talker read()s once per select wake-up and cuts the buffer at the
 *     first control character (nuts333.c:136,149,403-411), so pipelined lines vanish;
 *   - every client drains continuously: client sockets are blocking on the server side
 *     (nuts333.c:1192 sets O_NDELAY on the listen sockets only), a full receive queue
 *     stalls the whole talker inside write(2) (nuts333.c:1363);
 *   - clients leave by closing the socket, never with ".quit" (use-after-free at
 *     nuts333.c:1807-1809 -> 218-219);
 *   - login is a 3-stage FSM keyed on the prompts "Give me a name: " and
 *     "Give me a password: " (nuts333.c:309,1536) and ends with look()'s last line
 *     (nuts333.c:3998-4003).
 *
 * Threads: clients are partitioned over T worker threads, each with its own epoll
 * instance; the talker is single-threaded so the workers only have to keep up.
 *
 * Input: a workload spec on stdin (see parse_spec). Output: one JSON object on stdout.
 * Also: `loadgen --probe-write BYTES COUNT` measures the cost of one write(2) of BYTES
 * on a drained loopback TCP socket -- the per-recipient floor of the reference's
 * fan-out (one write per recipient per line, nuts333.c:1363).
 */
#define _GNU_SOURCE
#include <arpa/inet.h>
#include <errno.h>
#include <fcntl.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <pthread.h>
#include <sched.h>
#include <signal.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/epoll.h>
#include <sys/socket.h>
#include <sys/types.h>
#include <time.h>
#include <unistd.h>

#define MAX_THREADS 64
#define MAX_SERVERS 8
#define LBUF 4096
#define RBUF 65536

static const char LOOK_END_1[] = "has been set yet.*\n\r"; /* nuts333.c:4003; '*' = any bytes (ESC[0m with colour on) */
static const char NAME_PROMPT[] = "Give me a name: ";       /* nuts333.c:309  */
static const char PASS_PROMPT[] = "Give me a password: ";   /* nuts333.c:1536 */

enum cstate { ST_IDLE, ST_NAME, ST_PASS, ST_LOGIN_WAIT, ST_PRE, ST_READY, ST_RUN };

struct precmd { char *line; char *expect; };

struct client {
    int fd, idx, thread;
    char name[32], pass[32], host[64];
    int port;
    enum cstate st;
    /* phase buffer for prompt matching (login / placement) */
    char *acc; size_t acc_len, acc_cap;
    struct precmd *pre; int npre, ipre, cap_pre;
    struct sched { char **lines; int *len; int n, i, cap; } warm, timed, *cur;
    int awaiting_ack;
    uint64_t sent_ns;
    char lbuf[LBUF]; int llen;
    uint64_t rx_lines, rx_bytes, rx_acks;
};

struct worker {
    pthread_t tid; int id, epfd;
    struct client **cl; int ncl;
    int next_login, inflight, ready;
    uint64_t *lat; size_t nlat, caplat;
    int cpu;
    /* this thread over the timed window: CPU time it got and time it sat runnable on a run queue.  The workers
       busy-poll, so cpu/wall < 1 means the thread was descheduled (a neighbour on its core, or a cgroup throttle) */
    uint64_t cpu0_ns, cpu1_ns, rq0_ns, rq1_ns, w0_ns, w1_ns;   /* thread CPU, run-queue wait and wall clock at both ends of the timed window */
};

static struct client *g_clients; static int g_nclients, g_capclients;
static struct worker g_workers[MAX_THREADS]; static int g_nthreads = 4;
static int g_login_window = 4;
static double g_timeout_s = 300.0;
static int g_server_pids[MAX_SERVERS], g_nservers;
static uint64_t g_expect_lines, g_expect_warm;
static int g_cpus[256], g_ncpus;
static int g_verbose;
static double g_stall_s = 30.0;  /* no new line for this long in a closed-loop phase = a lost line: fail loudly */
static int g_quiet_ms = 40;   /* drain until every socket of the worker has been silent this long */
static int g_quickack = 1; /* re-arm TCP_QUICKACK after every read: the talker never sets TCP_NODELAY, so a second
                              small write to the same socket waits (Nagle) for our ACK, which the kernel would
                              otherwise delay by up to 40 ms -- that would time the delayed-ACK timer, not the talker */
static int g_spin = 1;   /* busy-poll in the timed phase: a sleeping receiver would make the talker pay a
                            cross-CPU wake-up inside every write(2), which measures the scheduler, not the talker */

static atomic_int g_phase;            /* 0 login, 1 drain, 2 run, 3 stop */
static atomic_int g_ready_workers;
static atomic_int g_drained_workers;
static atomic_ullong g_lines, g_bytes, g_acks;
static atomic_ullong g_t_end;
static atomic_int g_done, g_fail;
static pthread_barrier_t g_run_barrier, g_warm_barrier;

static uint64_t now_ns(void) {
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

static void die(const char *msg) { perror(msg); exit(2); }

static uint64_t thread_cpu_ns(void) {
    struct timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

/* second field of a schedstat file: ns spent runnable but waiting for a CPU (0 when the kernel keeps no schedstats) */
static uint64_t schedstat_run_delay(const char *path) {
    unsigned long long run = 0, delay = 0; FILE *fp = fopen(path, "r");
    if (!fp) return 0;
    if (fscanf(fp, "%llu %llu", &run, &delay) != 2) delay = 0;
    fclose(fp);
    return delay;
}

static void failf(struct client *c, const char *what) {
    fprintf(stderr, "loadgen: client %d (%s): %s (state %d, acc=%.*s)\n", c ? c->idx : -1,
            c ? c->name : "-", what, c ? (int)c->st : -1,
            c ? (int)(c->acc_len > 200 ? 200 : c->acc_len) : 0, c && c->acc ? c->acc : "");
    atomic_store(&g_fail, 1);
}

/* ------------------------------------------------------------------ spec parsing */
static struct client *client_at(int idx) {
    if (idx < 0 || idx >= g_nclients) { fprintf(stderr, "spec: bad client index %d\n", idx); exit(2); }
    return &g_clients[idx];
}

static void parse_spec(FILE *fp) {
    char *line = NULL; size_t cap = 0; ssize_t n;
    while ((n = getline(&line, &cap, fp)) > 0) {
        if (line[n - 1] == '\n') line[--n] = 0;
        if (!n || line[0] == '#') continue;
        if (!strncmp(line, "threads ", 8)) g_nthreads = atoi(line + 8);
        else if (!strncmp(line, "login_window ", 13)) g_login_window = atoi(line + 13);
        else if (!strncmp(line, "timeout_s ", 10)) g_timeout_s = atof(line + 10);
        else if (!strncmp(line, "verbose ", 8)) g_verbose = atoi(line + 8);
        else if (!strncmp(line, "drain_quiet_ms ", 15)) g_quiet_ms = atoi(line + 15);
        else if (!strncmp(line, "stall_s ", 8)) g_stall_s = atof(line + 8);
        else if (!strncmp(line, "spin ", 5)) g_spin = atoi(line + 5);
        else if (!strncmp(line, "quickack ", 9)) g_quickack = atoi(line + 9);
        else if (!strncmp(line, "expect_lines ", 13)) g_expect_lines = strtoull(line + 13, NULL, 10);
        else if (!strncmp(line, "expect_warm_lines ", 18)) g_expect_warm = strtoull(line + 18, NULL, 10);
        else if (!strncmp(line, "server_pid ", 11)) {
            if (g_nservers < MAX_SERVERS) g_server_pids[g_nservers++] = atoi(line + 11);
        } else if (!strncmp(line, "cpus ", 5)) {
            char *p = line + 5, *tok;
            while ((tok = strsep(&p, ",")) && g_ncpus < 256) if (*tok) g_cpus[g_ncpus++] = atoi(tok);
        } else if (!strncmp(line, "client ", 7)) {
            if (g_nclients == g_capclients) {
                g_capclients = g_capclients ? g_capclients * 2 : 64;
                g_clients = realloc(g_clients, sizeof(*g_clients) * (size_t)g_capclients);
            }
            struct client *c = &g_clients[g_nclients];
            memset(c, 0, sizeof(*c));
            c->idx = g_nclients; c->fd = -1;
            if (sscanf(line + 7, "%31s %31s %63s %d", c->name, c->pass, c->host, &c->port) != 4) {
                fprintf(stderr, "spec: bad client line: %s\n", line); exit(2);
            }
            g_nclients++;
        } else if (!strncmp(line, "pre ", 4)) {
            /* pre <idx> <expect>\t<line> */
            char *p = line + 4; int idx = (int)strtol(p, &p, 10);
            if (*p == ' ') p++;
            char *tab = strchr(p, '\t');
            if (!tab) { fprintf(stderr, "spec: pre needs TAB: %s\n", line); exit(2); }
            *tab = 0;
            struct client *c = client_at(idx);
            if (c->npre == c->cap_pre) {
                c->cap_pre = c->cap_pre ? c->cap_pre * 2 : 4;
                c->pre = realloc(c->pre, sizeof(*c->pre) * (size_t)c->cap_pre);
            }
            /* expect may use the two-character escapes \n and \r */
            char *e = strdup(p), *w = e;
            for (char *r = e; *r; r++) {
                if (r[0] == '\\' && r[1] == 'n') { *w++ = '\n'; r++; }
                else if (r[0] == '\\' && r[1] == 'r') { *w++ = '\r'; r++; }
                else *w++ = *r;
            }
            *w = 0;
            c->pre[c->npre].expect = e;
            c->pre[c->npre].line = strdup(tab + 1);
            c->npre++;
        } else if (!strncmp(line, "line ", 5) || !strncmp(line, "warm ", 5)) {
            /* line <idx> <text>: timed phase;  warm <idx> <text>: untimed warm-up phase before it */
            int is_warm = line[0] == 'w';
            char *p = line + 5; int idx = (int)strtol(p, &p, 10);
            if (*p == ' ') p++;
            struct client *c = client_at(idx);
            struct sched *sc = is_warm ? &c->warm : &c->timed;
            if (sc->n == sc->cap) {
                sc->cap = sc->cap ? sc->cap * 2 : 16;
                sc->lines = realloc(sc->lines, sizeof(char *) * (size_t)sc->cap);
                sc->len = realloc(sc->len, sizeof(int) * (size_t)sc->cap);
            }
            size_t l = strlen(p);
            char *t = malloc(l + 2); memcpy(t, p, l); t[l] = '\n'; t[l + 1] = 0;
            sc->lines[sc->n] = t; sc->len[sc->n] = (int)l + 1; sc->n++;
        } else { fprintf(stderr, "spec: unknown directive: %s\n", line); exit(2); }
    }
    free(line);
    if (g_nthreads < 1) g_nthreads = 1;
    if (g_nthreads > MAX_THREADS) g_nthreads = MAX_THREADS;
    if (g_nthreads > g_nclients && g_nclients > 0) g_nthreads = g_nclients;
}

/* ------------------------------------------------------------------ socket helpers */
static void send_all(struct client *c, const char *buf, int len) {
    /* one line == one send() == one segment (TCP_NODELAY is set) */
    int off = 0;
    while (off < len) {
        ssize_t w = send(c->fd, buf + off, (size_t)(len - off), MSG_NOSIGNAL);
        if (w < 0) {
            if (errno == EINTR) continue;
            if (errno == EAGAIN) { sched_yield(); continue; }
            failf(c, "send failed"); return;
        }
        off += (int)w;
    }
}

static void send_line(struct client *c, const char *s) {
    char tmp[1100]; int l = (int)strlen(s);
    if (l > 1000) l = 1000;
    memcpy(tmp, s, (size_t)l); tmp[l] = '\n';
    send_all(c, tmp, l + 1);
}

static void acc_reset(struct client *c) { c->acc_len = 0; }
static void acc_add(struct client *c, const char *b, size_t n) {
    if (c->acc_len + n + 1 > c->acc_cap) {
        c->acc_cap = (c->acc_len + n + 1) * 2;
        c->acc = realloc(c->acc, c->acc_cap);
    }
    memcpy(c->acc + c->acc_len, b, n); c->acc_len += n; c->acc[c->acc_len] = 0;
}
/* needle may contain one '*': the part before it must occur, and the part after it later on */
static int acc_has(struct client *c, const char *needle) {
    if (!c->acc_len) return 0;
    const char *star = strchr(needle, '*');
    if (!star) return memmem(c->acc, c->acc_len, needle, strlen(needle)) != NULL;
    const char *a = memmem(c->acc, c->acc_len, needle, (size_t)(star - needle));
    if (!a) return 0;
    a += star - needle;
    return memmem(a, c->acc_len - (size_t)(a - c->acc), star + 1, strlen(star + 1)) != NULL;
}

static void start_connect(struct worker *w, struct client *c) {
    struct sockaddr_in sa; memset(&sa, 0, sizeof(sa));
    sa.sin_family = AF_INET; sa.sin_port = htons((uint16_t)c->port);
    if (inet_pton(AF_INET, c->host, &sa.sin_addr) != 1) { failf(c, "bad host"); return; }
    c->fd = socket(AF_INET, SOCK_STREAM, 0);
    if (c->fd < 0) die("socket");
    int one = 1; setsockopt(c->fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
    int rcv = 1 << 20; setsockopt(c->fd, SOL_SOCKET, SO_RCVBUF, &rcv, sizeof(rcv));
    /* blocking connect on loopback is immediate; the talker's listen backlog is 10
       (nuts333.c:1189) so the login window must stay below that */
    if (connect(c->fd, (struct sockaddr *)&sa, sizeof(sa)) < 0) { failf(c, "connect failed"); return; }
    if (g_quickack) setsockopt(c->fd, IPPROTO_TCP, TCP_QUICKACK, &one, sizeof(one));
    int fl = fcntl(c->fd, F_GETFL, 0); fcntl(c->fd, F_SETFL, fl | O_NONBLOCK);
    struct epoll_event ev; ev.events = EPOLLIN; ev.data.ptr = c;
    if (epoll_ctl(w->epfd, EPOLL_CTL_ADD, c->fd, &ev) < 0) die("epoll_ctl");
    c->st = ST_NAME; acc_reset(c);
    w->inflight++;
}

/* advance the placement script; returns 1 when the client is READY */
static void pre_next(struct worker *w, struct client *c) {
    if (c->ipre < c->npre) {
        c->st = ST_PRE; acc_reset(c);
        send_line(c, c->pre[c->ipre].line);
    } else {
        c->st = ST_READY; acc_reset(c);
        w->inflight--; w->ready++;
    }
}

static void phase_bytes(struct worker *w, struct client *c, const char *b, size_t n) {
    switch (c->st) {
    case ST_NAME:
        acc_add(c, b, n);
        if (acc_has(c, NAME_PROMPT)) { c->st = ST_PASS; acc_reset(c); send_line(c, c->name); }
        else if (acc_has(c, "talker is full")) failf(c, "talker full (max_users)");
        break;
    case ST_PASS:
        acc_add(c, b, n);
        if (acc_has(c, PASS_PROMPT)) { c->st = ST_LOGIN_WAIT; acc_reset(c); send_line(c, c->pass); }
        else if (acc_has(c, NAME_PROMPT)) failf(c, "name rejected");
        break;
    case ST_LOGIN_WAIT:
        acc_add(c, b, n);
        if (acc_has(c, LOOK_END_1)) pre_next(w, c);
        else if (acc_has(c, "Incorrect login") || acc_has(c, "confirm password"))
            failf(c, "login rejected (account not provisioned?)");
        break;
    case ST_PRE:
        acc_add(c, b, n);
        if (acc_has(c, c->pre[c->ipre].expect)) { c->ipre++; pre_next(w, c); }
        else if (acc_has(c, "Unknown command") || acc_has(c, "not adjoined") || acc_has(c, "no such room"))
            failf(c, "placement command rejected");
        break;
    default: break; /* READY: sign-on / movement chatter from other clients is discarded */
    }
}

/* ------------------------------------------------------------------ timed phase */
static inline int is_ack_line(const char *s, int len) {
    /* every acknowledgement the three commands produce starts with "You "
       (nuts333.c:4094,4119,4176), possibly preceded by '\r' left over from the previous
       "\n\r" and by ANSI sequences when the recipient has colour on */
    int i = 0;
    for (;;) {
        while (i < len && s[i] == '\r') i++;
        if (i + 1 < len && s[i] == 27 && s[i + 1] == '[') {
            i += 2;
            while (i < len && s[i] != 'm') i++;
            if (i < len) i++;
            continue;
        }
        break;
    }
    return len - i >= 4 && s[i] == 'Y' && s[i + 1] == 'o' && s[i + 2] == 'u' && s[i + 3] == ' ';
}

static void send_next(struct client *c) {
    struct sched *sc = c->cur;
    if (sc && sc->i < sc->n) {
        c->awaiting_ack = 1;
        c->sent_ns = now_ns();
        send_all(c, sc->lines[sc->i], sc->len[sc->i]);
        sc->i++;
    }
}

static inline void on_line(struct worker *w, struct client *c, const char *s, int len) {
    c->rx_lines++;
    if (c->awaiting_ack && is_ack_line(s, len)) {
        c->awaiting_ack = 0; c->rx_acks++;
        if (w->nlat == w->caplat) {
            w->caplat = w->caplat ? w->caplat * 2 : 4096;
            w->lat = realloc(w->lat, sizeof(uint64_t) * w->caplat);
        }
        w->lat[w->nlat++] = now_ns() - c->sent_ns;
        send_next(c);
    }
}

static void run_bytes(struct worker *w, struct client *c, const char *b, size_t n) {
    c->rx_bytes += n;
    const char *p = b, *end = b + n;
    if (c->llen) {
        const char *nl = memchr(p, '\n', (size_t)(end - p));
        size_t take = nl ? (size_t)(nl - p) : (size_t)(end - p);
        if (c->llen + (int)take > LBUF) take = (size_t)(LBUF - c->llen);
        memcpy(c->lbuf + c->llen, p, take); c->llen += (int)take;
        if (!nl) return;
        on_line(w, c, c->lbuf, c->llen); c->llen = 0;
        p = nl + 1;
    }
    while (p < end) {
        const char *nl = memchr(p, '\n', (size_t)(end - p));
        if (!nl) {
            size_t rest = (size_t)(end - p);
            if (rest > LBUF) rest = LBUF;
            memcpy(c->lbuf, p, rest); c->llen = (int)rest;
            return;
        }
        on_line(w, c, p, (int)(nl - p));
        p = nl + 1;
    }
}

/* ------------------------------------------------------------------ worker */
static int drain_client(struct worker *w, struct client *c, char *rbuf, int running) {
    /* returns bytes read in total; -1 on close */
    int total = 0;
    for (;;) {
        ssize_t r = recv(c->fd, rbuf, RBUF, 0);
        if (r > 0) {
            total += (int)r;
            if (running) run_bytes(w, c, rbuf, (size_t)r); else phase_bytes(w, c, rbuf, (size_t)r);
            if (r < RBUF) break;
            continue;
        }
        if (r == 0) { failf(c, "server closed the connection"); return -1; }
        if (errno == EINTR) continue;
        if (errno == EAGAIN || errno == EWOULDBLOCK) break;
        failf(c, "recv failed"); return -1;
    }
    if (g_quickack && total) { int one = 1; setsockopt(c->fd, IPPROTO_TCP, TCP_QUICKACK, &one, sizeof(one)); }
    return total;
}

/* Everything the logins / placement / warm-up caused is already written by the talker (it writes
   the acting user's own output last), but may still be in flight -- over a netlink it can sit
   behind Nagle + a delayed ACK for tens of ms.  Read until all our sockets stay silent. */
static void drain_until_quiet(struct worker *w, char *rbuf) {
    uint64_t last = now_ns();
    while (now_ns() - last < (uint64_t)g_quiet_ms * 1000000ull && !atomic_load(&g_fail)) {
        int got = 0;
        for (int i = 0; i < w->ncl; i++) if (drain_client(w, w->cl[i], rbuf, 0) > 0) got = 1;
        if (got) last = now_ns();
        usleep(1000);
    }
}

static void *worker_main(void *arg) {
    struct worker *w = arg;
    char *rbuf = malloc(RBUF);
    struct epoll_event evs[256];
    if (w->cpu >= 0) {
        cpu_set_t set; CPU_ZERO(&set); CPU_SET(w->cpu, &set);
        pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
    }
    int announced_ready = 0, announced_drained = 0, started_run = 0, started_warm = 0;
    uint64_t pub_lines = 0, pub_bytes = 0, pub_acks = 0;
    while (!atomic_load(&g_fail)) {
        int phase = atomic_load(&g_phase);
        if (phase == 3) break;
        if (phase == 0) {
            while (w->inflight < g_login_window && w->next_login < w->ncl)
                start_connect(w, w->cl[w->next_login++]);
            if (!announced_ready && w->ready == w->ncl) {
                announced_ready = 1; atomic_fetch_add(&g_ready_workers, 1);
            }
        }
        if (phase == 4 && !started_warm) {
            /* untimed warm-up: same closed-loop machinery, its own schedule and counters */
            drain_until_quiet(w, rbuf);
            for (int i = 0; i < w->ncl; i++) {
                struct client *c = w->cl[i];
                c->st = ST_RUN; c->llen = 0; c->rx_lines = c->rx_bytes = c->rx_acks = 0; c->cur = &c->warm;
            }
            started_warm = 1; started_run = 1; pub_lines = pub_bytes = pub_acks = 0;
            /* nobody may send before EVERY worker has stopped discarding: a broadcast reaching a
               client whose worker is still draining would be thrown away and the count never met */
            pthread_barrier_wait(&g_warm_barrier);
            for (int i = 0; i < w->ncl; i++) send_next(w->cl[i]);
            continue;
        }
        if (phase == 1 && !announced_drained) {
            started_run = 0;
            /* every broadcast caused by logins/placement is already queued on our sockets
               (the talker writes the acting user's own look() output last); empty them */
            drain_until_quiet(w, rbuf);
            for (int i = 0; i < w->ncl; i++) {
                struct client *c = w->cl[i];
                c->st = ST_RUN; c->llen = 0; c->rx_lines = c->rx_bytes = c->rx_acks = 0; c->cur = &c->timed;
                c->awaiting_ack = 0;
            }
            pub_lines = pub_bytes = pub_acks = 0; w->nlat = 0;
            announced_drained = 1; atomic_fetch_add(&g_drained_workers, 1);
            pthread_barrier_wait(&g_run_barrier);   /* main records t0 then joins */
            pthread_barrier_wait(&g_run_barrier);
            started_run = 1;
            w->cpu0_ns = thread_cpu_ns(); w->rq0_ns = schedstat_run_delay("/proc/thread-self/schedstat"); w->w0_ns = now_ns();
            for (int i = 0; i < w->ncl; i++) send_next(w->cl[i]);
            continue;
        }
        int n = epoll_wait(w->epfd, evs, 256, started_run ? (g_spin ? 0 : 5) : 20);
        if (n < 0) { if (errno == EINTR) continue; die("epoll_wait"); }
        for (int i = 0; i < n; i++) {
            struct client *c = evs[i].data.ptr;
            if (drain_client(w, c, rbuf, started_run) < 0) break;
        }
        if (started_run) {
            uint64_t l = 0, b = 0, a = 0;
            for (int i = 0; i < w->ncl; i++) { l += w->cl[i]->rx_lines; b += w->cl[i]->rx_bytes; a += w->cl[i]->rx_acks; }
            if (l != pub_lines || b != pub_bytes) {
                uint64_t tot = atomic_fetch_add(&g_lines, l - pub_lines) + (l - pub_lines);
                atomic_fetch_add(&g_bytes, b - pub_bytes);
                atomic_fetch_add(&g_acks, a - pub_acks);
                pub_lines = l; pub_bytes = b; pub_acks = a;
                if (phase == 2 && tot >= g_expect_lines && !atomic_exchange(&g_done, 1))
                    atomic_store(&g_t_end, now_ns());
            }
            if (!w->cpu1_ns && atomic_load(&g_done)) {
                w->cpu1_ns = thread_cpu_ns(); w->rq1_ns = schedstat_run_delay("/proc/thread-self/schedstat"); w->w1_ns = now_ns();
            }
        }
    }
    /* a worker that got no loop turn between g_done and the stop (descheduled through the grace period, or asleep in a
       5 ms epoll_wait with spin 0) still reports its counters: sampled here, over the longer window it really covers */
    if (w->cpu0_ns && !w->cpu1_ns) {
        w->cpu1_ns = thread_cpu_ns(); w->rq1_ns = schedstat_run_delay("/proc/thread-self/schedstat"); w->w1_ns = now_ns();
    }
    free(rbuf);
    return NULL;
}

/* ------------------------------------------------------------------ /proc sampling */
struct cpu_sample { double utime_s, stime_s; uint64_t sched_ns, rq_ns, vcsw, ivcsw, syscr, syscw, wchar; };

static void sample_pid(int pid, struct cpu_sample *s) {
    char path[64], buf[1024]; memset(s, 0, sizeof(*s));
    snprintf(path, sizeof(path), "/proc/%d/stat", pid);
    FILE *fp = fopen(path, "r");
    if (fp) {
        if (fgets(buf, sizeof(buf), fp)) {
            char *p = strrchr(buf, ')');   /* comm may contain spaces */
            unsigned long ut = 0, st = 0;
            if (p && sscanf(p + 2, "%*c %*d %*d %*d %*d %*d %*u %*u %*u %*u %*u %lu %lu", &ut, &st) == 2) {
                double hz = (double)sysconf(_SC_CLK_TCK);
                s->utime_s = (double)ut / hz; s->stime_s = (double)st / hz;
            }
        }
        fclose(fp);
    }
    snprintf(path, sizeof(path), "/proc/%d/schedstat", pid);
    fp = fopen(path, "r");
    if (fp) {
        unsigned long long ns = 0, rq = 0;
        int got = fscanf(fp, "%llu %llu", &ns, &rq);
        if (got >= 1) s->sched_ns = ns;
        if (got >= 2) s->rq_ns = rq;       /* runnable, waiting for the CPU: contention on the talker's core or a throttle */
        fclose(fp);
    }
    /* voluntary switches = the talker went to sleep (select with nothing to read, or write(2) into a full socket);
       involuntary = something else took its core */
    snprintf(path, sizeof(path), "/proc/%d/status", pid);
    fp = fopen(path, "r");
    if (fp) {
        while (fgets(buf, sizeof(buf), fp)) {
            unsigned long long v;
            if (sscanf(buf, "voluntary_ctxt_switches: %llu", &v) == 1) s->vcsw = v;
            else if (sscanf(buf, "nonvoluntary_ctxt_switches: %llu", &v) == 1) s->ivcsw = v;
        }
        fclose(fp);
    }
    /* exact read/write system-call counts of the talker: /proc/<pid>/io (same uid) */
    snprintf(path, sizeof(path), "/proc/%d/io", pid);
    fp = fopen(path, "r");
    if (fp) {
        char key[32]; unsigned long long v;
        while (fscanf(fp, "%31s %llu", key, &v) == 2) {
            if (!strcmp(key, "syscr:")) s->syscr = v;
            else if (!strcmp(key, "syscw:")) s->syscw = v;
            else if (!strcmp(key, "wchar:")) s->wchar = v;
        }
        fclose(fp);
    }
}

static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b; return x < y ? -1 : x > y;
}

/* ------------------------------------------------------------------ write(2) probe */
struct probe_arg { int fd; };
static void *probe_reader(void *arg) {
    struct probe_arg *pa = arg; char *buf = malloc(RBUF);
    while (recv(pa->fd, buf, RBUF, 0) > 0) {}
    free(buf); return NULL;
}

static int probe_write(int bytes, long count) {
    int ls = socket(AF_INET, SOCK_STREAM, 0); if (ls < 0) die("socket");
    struct sockaddr_in sa; memset(&sa, 0, sizeof(sa));
    sa.sin_family = AF_INET; sa.sin_addr.s_addr = htonl(INADDR_LOOPBACK); sa.sin_port = 0;
    if (bind(ls, (struct sockaddr *)&sa, sizeof(sa)) < 0) die("bind");
    socklen_t sl = sizeof(sa); getsockname(ls, (struct sockaddr *)&sa, &sl);
    listen(ls, 1);
    int cfd = socket(AF_INET, SOCK_STREAM, 0);
    if (connect(cfd, (struct sockaddr *)&sa, sizeof(sa)) < 0) die("connect");
    int sfd = accept(ls, NULL, NULL); if (sfd < 0) die("accept");
    /* the talker does not set TCP_NODELAY on its sockets; neither do we on the writer */
    int rcv = 1 << 20; setsockopt(cfd, SOL_SOCKET, SO_RCVBUF, &rcv, sizeof(rcv));
    struct probe_arg pa = { cfd }; pthread_t rt;
    pthread_create(&rt, NULL, probe_reader, &pa);
    char *buf = malloc((size_t)bytes); memset(buf, 'x', (size_t)bytes);
    for (int i = 0; i < 2000; i++) if (write(sfd, buf, (size_t)bytes) < 0) die("write");
    struct timespec c0, c1; uint64_t t0 = now_ns();
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &c0);
    for (long i = 0; i < count; i++) if (write(sfd, buf, (size_t)bytes) < 0) die("write");
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &c1);
    uint64_t t1 = now_ns();
    double cpu_ns = (double)(c1.tv_sec - c0.tv_sec) * 1e9 + (double)(c1.tv_nsec - c0.tv_nsec);
    printf("{\"probe\":\"write\",\"bytes\":%d,\"count\":%ld,\"wall_ns_per_write\":%.1f,\"cpu_ns_per_write\":%.1f}\n",
           bytes, count, (double)(t1 - t0) / (double)count, cpu_ns / (double)count);
    close(sfd); pthread_join(rt, NULL); close(cfd); close(ls); free(buf);
    return 0;
}


/* ------------------------------------------------------------------ per-input-line syscall probe
 *
 * The floor of the reference's per-input-line work with ZERO user-space work:
 *
 *     select(FD_SETSIZE, &readmask, ...)        nuts333.c:94     one per input line, O(nfds) scan
 *     read(sender, buf, ...)                    nuts333.c:136    one per input line
 *     write(sender, ack) ; write(recipient_i)   nuts333.c:1363   1 + K per input line (say/shout: the
 *                                                                sender's "You ...:" line goes first)
 *
 * One "talker" thread does exactly that and nothing else over K+1 connected loopback pairs; R reader
 * threads play the synthetic clients the way the load generator's workers do (own epoll, busy-poll,
 * TCP_QUICKACK re-armed).  Client 0 is the sender: it sends the next LINE-byte input line when its ack
 * arrives -- the load generator's closed loop.
 *
 *   selread=0   write-only leg: the K+1 writes alone (round 1's figure, kept for comparison)
 *   selread=1   full leg: select + read + K+1 writes
 *   open=1      open loop: the sender keeps DEPTH lines in flight so select() never sleeps and the
 *               talker thread never waits for a receiver: its WALL-clock rate is a demonstrated rate,
 *               not a CPU-time extrapolation.
 */
#include <sys/resource.h>
#include <sys/select.h>

#define PROBE_MAX_READERS 16
struct lp_reader { int epfd, cpu, sender_cfd, line, ack_bytes, feed; atomic_int *stop; atomic_ullong rx; pthread_t tid; };

static void *lp_reader_main(void *arg) {
    struct lp_reader *rd = arg; char *buf = malloc(RBUF); struct epoll_event evs[256];
    char line[1024]; memset(line, 'i', sizeof(line));
    if (rd->cpu >= 0) { cpu_set_t set; CPU_ZERO(&set); CPU_SET(rd->cpu, &set); pthread_setaffinity_np(pthread_self(), sizeof(set), &set); }
    line[rd->line - 1] = '\n';
    unsigned long long sender_rx = 0, sender_lines = 0; int ackbytes = rd->ack_bytes;   /* every ack is one BYTES-long line */
    while (!atomic_load(rd->stop)) {
        int n = epoll_wait(rd->epfd, evs, 256, 0);
        for (int i = 0; i < n; i++) {
            int fd = evs[i].data.fd; ssize_t r; unsigned long long got = 0;
            while ((r = recv(fd, buf, RBUF, MSG_DONTWAIT)) > 0) got += (unsigned long long)r;
            int one = 1; setsockopt(fd, IPPROTO_TCP, TCP_QUICKACK, &one, sizeof(one));
            atomic_fetch_add(&rd->rx, got);
            if (fd == rd->sender_cfd) sender_rx += got;
        }
        /* One new input line per ack received, each its own segment (TCP_NODELAY on the client side) -- but only
           when the talker thread reads them (feed = the select+read legs).  In the write-only leg nobody reads the
           sender's socket: lines sent there would pile up until the buffers fill and this thread blocks in send(),
           stops draining, and the talker thread waits for bytes that never arrive (hung for
           rounds x 60 B above the socket buffers).  Never block here: EAGAIN = try again on the next turn. */
        while (rd->feed && rd->sender_cfd >= 0 && sender_lines < sender_rx / (unsigned long long)ackbytes) {
            if (send(rd->sender_cfd, line, (size_t)rd->line, MSG_NOSIGNAL | MSG_DONTWAIT) < 0) break;
            sender_lines++;
        }
    }
    free(buf); return NULL;
}

static int probe_line(int bytes, int k, long rounds, int selread, int open_loop, int nreaders, int wcpu, const int *rcpus, int nrcpus) {
    const int LINE = 60, DEPTH = 32;        /* input line: 54-byte payload + ".shout" / newline, about 60 bytes */
    if (nreaders < 1) nreaders = 1;
    if (nreaders > PROBE_MAX_READERS) nreaders = PROBE_MAX_READERS;
    if (bytes < 2 || bytes > 1000 || k < 0 || k + 8 >= FD_SETSIZE) { fprintf(stderr, "probe-line: bad sizes\n"); return 2; }
    struct rlimit rl; if (!getrlimit(RLIMIT_NOFILE, &rl)) { rl.rlim_cur = rl.rlim_max; setrlimit(RLIMIT_NOFILE, &rl); }
    int ls = socket(AF_INET, SOCK_STREAM, 0); if (ls < 0) die("socket");
    struct sockaddr_in sa; memset(&sa, 0, sizeof(sa));
    sa.sin_family = AF_INET; sa.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
    if (bind(ls, (struct sockaddr *)&sa, sizeof(sa)) < 0) die("bind");
    socklen_t sl = sizeof(sa); getsockname(ls, (struct sockaddr *)&sa, &sl);
    listen(ls, 16);
    int nsock = k + 1;                       /* [0] = the sender, [1..k] = recipients */
    int *wfd = calloc((size_t)nsock, sizeof(int));
    atomic_int stop; atomic_init(&stop, 0);
    struct lp_reader rd[PROBE_MAX_READERS];
    for (int r = 0; r < nreaders; r++) {
        rd[r].epfd = epoll_create1(0); rd[r].cpu = nrcpus ? rcpus[r % nrcpus] : -1; rd[r].sender_cfd = -1;
        rd[r].line = LINE; rd[r].ack_bytes = bytes; rd[r].feed = selread; rd[r].stop = &stop; atomic_init(&rd[r].rx, 0);
    }
    int sender_cfd = -1;
    for (int i = 0; i < nsock; i++) {
        int cfd = socket(AF_INET, SOCK_STREAM, 0); if (cfd < 0) die("socket");
        int one = 1; setsockopt(cfd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
        int rcv = 1 << 20; setsockopt(cfd, SOL_SOCKET, SO_RCVBUF, &rcv, sizeof(rcv));
        if (connect(cfd, (struct sockaddr *)&sa, sizeof(sa)) < 0) die("connect");
        int a = accept(ls, NULL, NULL); if (a < 0) die("accept");
        /* the talker side must stay below FD_SETSIZE for select(); park the client side high */
        int hi = fcntl(cfd, F_DUPFD, 2048); if (hi < 0) die("F_DUPFD (RLIMIT_NOFILE too low)");
        close(cfd); cfd = hi;
        if (a >= FD_SETSIZE) { fprintf(stderr, "probe-line: talker-side fd %d >= FD_SETSIZE\n", a); return 2; }
        wfd[i] = a;                          /* blocking, no TCP_NODELAY: as the talker leaves its sockets */
        setsockopt(cfd, IPPROTO_TCP, TCP_QUICKACK, &one, sizeof(one));
        struct epoll_event ev; ev.events = EPOLLIN; ev.data.fd = cfd;
        struct lp_reader *r = &rd[i % nreaders];
        epoll_ctl(r->epfd, EPOLL_CTL_ADD, cfd, &ev);
        if (i == 0) { sender_cfd = cfd; r->sender_cfd = cfd; }
    }
    for (int r = 0; r < nreaders; r++) pthread_create(&rd[r].tid, NULL, lp_reader_main, &rd[r]);
    if (wcpu >= 0) { cpu_set_t set; CPU_ZERO(&set); CPU_SET(wcpu, &set); pthread_setaffinity_np(pthread_self(), sizeof(set), &set); }
    char *buf = malloc((size_t)bytes); memset(buf, 'x', (size_t)bytes); buf[bytes - 2] = '\n'; buf[bytes - 1] = '\r';
    char in[1024], first[1024]; memset(first, 'i', sizeof(first)); first[LINE - 1] = '\n';
    fd_set proto, mask; FD_ZERO(&proto); FD_SET(ls, &proto);
    for (int i = 0; i < nsock; i++) FD_SET(wfd[i], &proto);
    /* prime: in the closed loop one line in flight, in the open loop DEPTH of them; every ack then triggers one more */
    int prime = selread ? (open_loop ? DEPTH : 1) : 0;
    for (int i = 0; i < prime; i++) if (send(sender_cfd, first, (size_t)LINE, MSG_NOSIGNAL) < 0) die("prime send");
    const long WARM = 20;
    unsigned long long want = 0; struct timespec s0, s1, w1; uint64_t t0 = 0; double sel_ns = 0, wr_ns = 0;
#define TS_NS(a, b) ((double)((b).tv_sec - (a).tv_sec) * 1e9 + (double)((b).tv_nsec - (a).tv_nsec))
    for (long r = -WARM; r < rounds; r++) {
        if (r == 0) { t0 = now_ns(); sel_ns = wr_ns = 0; }
        /* CPU time is taken per section so that neither the write-only leg's wait below nor a sleep in select()
           is charged; the three clock calls per round (~0.1 us each, a real system call for this clock) are
           ours, not the talker's, and about one of them lands inside each section: the peak is understated by
           that much, which matters only for small K */
        clock_gettime(CLOCK_THREAD_CPUTIME_ID, &s0);
        if (selread) {
            mask = proto;
            if (select(FD_SETSIZE, &mask, NULL, NULL, NULL) < 0) die("select");
            ssize_t got = read(wfd[0], in, (size_t)LINE);     /* lines are LINE bytes each: exactly one */
            if (got <= 0) die("read");
        }
        clock_gettime(CLOCK_THREAD_CPUTIME_ID, &s1);
        for (int i = 0; i < nsock; i++) if (write(wfd[i], buf, (size_t)bytes) < 0) die("write");
        clock_gettime(CLOCK_THREAD_CPUTIME_ID, &w1);
        if (selread) sel_ns += TS_NS(s0, s1);
        wr_ns += TS_NS(s1, w1);
        want += (unsigned long long)bytes * (unsigned long long)nsock;
        if (!selread) {
            /* write-only leg keeps round 1's closed loop: wait until the readers hold every byte */
            unsigned long long have;
            do { have = 0; for (int q = 0; q < nreaders; q++) have += atomic_load(&rd[q].rx); } while (have < want);
        }
    }
    uint64_t t1 = now_ns();
    /* let the readers finish so the byte count can be checked */
    unsigned long long total = (unsigned long long)(rounds + WARM) * (unsigned long long)bytes * (unsigned long long)nsock, have = 0;
    uint64_t dl = now_ns() + 5000000000ull;
    do { have = 0; for (int q = 0; q < nreaders; q++) have += atomic_load(&rd[q].rx); } while (have < total && now_ns() < dl);
    atomic_store(&stop, 1);
    for (int r = 0; r < nreaders; r++) pthread_join(rd[r].tid, NULL);
    double cpu_ns = sel_ns + wr_ns;
    double wall_ns = (double)(t1 - t0), nw = (double)rounds * (double)nsock;
    printf("{\"probe\":\"line\",\"mode\":\"%s\",\"select_read\":%d,\"bytes\":%d,\"recipients\":%d,\"writes_per_line\":%d,"
           "\"input_line_bytes\":%d,\"select_nfds\":%d,\"rounds\":%ld,\"readers\":%d,"
           "\"cpu_ns_per_line\":%.1f,\"wall_ns_per_line\":%.1f,\"cpu_ns_select_read_per_line\":%.1f,"
           "\"cpu_ns_per_write\":%.1f,\"cpu_ns_per_written_line\":%.1f,\"wall_ns_per_written_line\":%.1f,"
           "\"written_lines_per_s_cpu\":%.0f,\"written_lines_per_s_wall\":%.0f,\"bytes_ok\":%s}\n",
           open_loop ? "open" : "closed", selread, bytes, k, nsock, selread ? LINE : 0, selread ? FD_SETSIZE : 0, rounds, nreaders,
           cpu_ns / (double)rounds, wall_ns / (double)rounds, sel_ns / (double)rounds,
           wr_ns / nw, cpu_ns / nw, wall_ns / nw, nw / (cpu_ns / 1e9), nw / (wall_ns / 1e9), have >= total ? "true" : "false");
    return have >= total ? 0 : 1;
}

/* round 1's CLI: `--probe-fanout BYTES K ROUNDS [WCPU [RCPU]]` = the closed-loop write-only leg over K sockets */
static int probe_fanout(int bytes, int k, long rounds, int wcpu, int rcpu) {
    return probe_line(bytes, k - 1, rounds, 0, 0, 1, wcpu, &rcpu, rcpu >= 0 ? 1 : 0);
}

/* ------------------------------------------------------------------ main */
int main(int argc, char **argv) {
    signal(SIGPIPE, SIG_IGN);
    if (argc >= 4 && !strcmp(argv[1], "--probe-write")) return probe_write(atoi(argv[2]), atol(argv[3]));
    if (argc >= 5 && !strcmp(argv[1], "--probe-line")) {
        /* --probe-line BYTES RECIPIENTS ROUNDS [selread=0|1] [open=0|1] [readers] [wcpu] [rcpu,rcpu,...] */
        int rc[PROBE_MAX_READERS], nrc = 0;
        if (argc > 9) { char *p = argv[9], *tok; while ((tok = strsep(&p, ",")) && nrc < PROBE_MAX_READERS) if (*tok) rc[nrc++] = atoi(tok); }
        return probe_line(atoi(argv[2]), atoi(argv[3]), atol(argv[4]), argc > 5 ? atoi(argv[5]) : 1, argc > 6 ? atoi(argv[6]) : 0,
                          argc > 7 ? atoi(argv[7]) : 2, argc > 8 ? atoi(argv[8]) : -1, rc, nrc);
    }
    if (argc >= 5 && !strcmp(argv[1], "--probe-fanout"))
        return probe_fanout(atoi(argv[2]), atoi(argv[3]), atol(argv[4]), argc > 5 ? atoi(argv[5]) : -1, argc > 6 ? atoi(argv[6]) : -1);
    FILE *fp = stdin;
    if (argc >= 2) { fp = fopen(argv[1], "r"); if (!fp) die("open spec"); }
    parse_spec(fp);
    if (!g_nclients) { fprintf(stderr, "spec: no clients\n"); return 2; }

    for (int t = 0; t < g_nthreads; t++) {
        struct worker *w = &g_workers[t];
        w->id = t; w->epfd = epoll_create1(0); if (w->epfd < 0) die("epoll_create1");
        w->cl = calloc((size_t)g_nclients, sizeof(*w->cl));
        w->cpu = g_ncpus ? g_cpus[t % g_ncpus] : -1;
    }
    for (int i = 0; i < g_nclients; i++) {
        struct worker *w = &g_workers[i % g_nthreads];
        g_clients[i].thread = w->id; w->cl[w->ncl++] = &g_clients[i];
    }
    pthread_barrier_init(&g_run_barrier, NULL, (unsigned)g_nthreads + 1);
    pthread_barrier_init(&g_warm_barrier, NULL, (unsigned)g_nthreads);
    uint64_t t_login0 = now_ns();
    for (int t = 0; t < g_nthreads; t++) pthread_create(&g_workers[t].tid, NULL, worker_main, &g_workers[t]);

    uint64_t deadline = now_ns() + (uint64_t)(g_timeout_s * 1e9);
    while (atomic_load(&g_ready_workers) < g_nthreads && !atomic_load(&g_fail)) {
        if (now_ns() > deadline) { fprintf(stderr, "loadgen: login phase timed out\n"); atomic_store(&g_fail, 1); }
        usleep(2000);
    }
    uint64_t t_login1 = now_ns();
    struct cpu_sample s0[MAX_SERVERS], s1[MAX_SERVERS], s2[MAX_SERVERS];
    uint64_t t0 = 0, t1 = 0; int timed_out = 0;
    const uint64_t GAP_NS = 5000000ull;      /* 5 ms without a single line anywhere */
    uint64_t gap_seen = 0, gap_seen_at = 0, gap_idle_poll = 0, gap_max = 0, gap_total = 0; unsigned gap_n = 0;
    double warm_s = 0.0;
    if (!atomic_load(&g_fail) && g_expect_warm) {
        uint64_t w0 = now_ns();
        atomic_store(&g_phase, 4);
        uint64_t seen = 0, seen_at = now_ns();
        while (atomic_load(&g_lines) < g_expect_warm && !atomic_load(&g_fail)) {
            uint64_t l = atomic_load(&g_lines), n = now_ns();
            if (l != seen) { seen = l; seen_at = n; }
            if (n > deadline || (double)(n - seen_at) > g_stall_s * 1e9) {
                fprintf(stderr, "loadgen: warm-up %s at %llu/%llu lines\n", n > deadline ? "timed out" : "stalled",
                        (unsigned long long)l, (unsigned long long)g_expect_warm);
                atomic_store(&g_fail, 1);
            }
            usleep(500);
        }
        warm_s = (double)(now_ns() - w0) / 1e9;
    }
    if (!atomic_load(&g_fail)) {
        atomic_store(&g_phase, 1);
        while (atomic_load(&g_drained_workers) < g_nthreads && !atomic_load(&g_fail)) usleep(1000);
        pthread_barrier_wait(&g_run_barrier);
        atomic_store(&g_lines, 0); atomic_store(&g_bytes, 0); atomic_store(&g_acks, 0);   /* workers are parked */
        for (int i = 0; i < g_nservers; i++) sample_pid(g_server_pids[i], &s0[i]);
        t0 = now_ns();
        atomic_store(&g_phase, 2);
        pthread_barrier_wait(&g_run_barrier);
        uint64_t last_prog = t0, last_lines = 0;
        gap_seen_at = t0;
        while (!atomic_load(&g_done) && !atomic_load(&g_fail)) {
            usleep(500);
            uint64_t n = now_ns();
            if (n > deadline) { timed_out = 1; break; }
            {
                /* progress gaps: intervals in which NO line reached ANY client.  A gap is counted from the moment
                   progress was last seen to the last poll that still saw none, so a late wake-up of this thread
                   cannot invent one.  In a saturating single-talker run lines arrive every 1-2 us: a gap of
                   milliseconds means the talker (or the one sender that feeds it) was not running. */
                uint64_t l = atomic_load(&g_lines);
                if (l != gap_seen) {
                    if (gap_idle_poll > gap_seen_at) {
                        uint64_t gap = gap_idle_poll - gap_seen_at;
                        if (gap > gap_max) gap_max = gap;
                        if (gap >= GAP_NS) { gap_n++; gap_total += gap; }
                    }
                    gap_seen = l; gap_seen_at = n; gap_idle_poll = 0;
                } else gap_idle_poll = n;
                if ((double)(n - gap_seen_at) > g_stall_s * 1e9) {
                    fprintf(stderr, "loadgen: stalled at %llu/%llu lines for %.0f s\n", (unsigned long long)l,
                            (unsigned long long)g_expect_lines, g_stall_s);
                    timed_out = 1; break;
                }
            }
            if (g_verbose && n - last_prog > 1000000000ull) {
                uint64_t l = atomic_load(&g_lines);
                fprintf(stderr, "loadgen: %llu/%llu lines (+%llu)\n", (unsigned long long)l,
                        (unsigned long long)g_expect_lines, (unsigned long long)(l - last_lines));
                last_lines = l; last_prog = n;
            }
        }
        t1 = atomic_load(&g_done) ? atomic_load(&g_t_end) : now_ns();
        for (int i = 0; i < g_nservers; i++) sample_pid(g_server_pids[i], &s1[i]);
        /* grace: anything beyond the expected count is an error worth seeing */
        if (!timed_out) usleep(50000);
        /* system-call COUNTS are read after the grace period: the talker may still owe a write that no client
           waits for (the PRM frame that follows a remote user's command, nuts333.c:2181), and with every sender
           finished nothing else can add to them.  CPU time stays as sampled at the end of the timed window. */
        for (int i = 0; i < g_nservers; i++) {
            sample_pid(g_server_pids[i], &s2[i]);
            s1[i].syscr = s2[i].syscr; s1[i].syscw = s2[i].syscw; s1[i].wchar = s2[i].wchar;
        }
    }
    int failed = atomic_load(&g_fail);
    if (failed) {
        /* workers may be parked on the run barrier; do not try to join them */
        printf("{\"ok\":false,\"error\":\"see stderr\",\"clients\":%d}\n", g_nclients);
        fflush(stdout);
        _exit(1);
    }
    atomic_store(&g_phase, 3);
    for (int t = 0; t < g_nthreads; t++) pthread_join(g_workers[t].tid, NULL);
    /* leave by closing the socket, never .quit */
    for (int i = 0; i < g_nclients; i++) if (g_clients[i].fd >= 0) close(g_clients[i].fd);

    uint64_t lines = 0, bytes = 0, acks = 0, sent = 0, planned = 0;
    for (int i = 0; i < g_nclients; i++) {
        lines += g_clients[i].rx_lines; bytes += g_clients[i].rx_bytes; acks += g_clients[i].rx_acks;
        sent += (uint64_t)g_clients[i].timed.i; planned += (uint64_t)g_clients[i].timed.n;
    }
    size_t nlat = 0;
    for (int t = 0; t < g_nthreads; t++) nlat += g_workers[t].nlat;
    uint64_t *lat = malloc(sizeof(uint64_t) * (nlat ? nlat : 1)); size_t k = 0;
    for (int t = 0; t < g_nthreads; t++) {
        if (g_workers[t].nlat) memcpy(lat + k, g_workers[t].lat, sizeof(uint64_t) * g_workers[t].nlat);
        k += g_workers[t].nlat;
    }
    qsort(lat, nlat, sizeof(uint64_t), cmp_u64);
    double lat_mean = 0; for (size_t i = 0; i < nlat; i++) lat_mean += (double)lat[i];
    if (nlat) lat_mean /= (double)nlat;

    double wall = (double)(t1 - t0) / 1e9;
    printf("{\"ok\":%s,\"timed_out\":%d,\"clients\":%d,\"threads\":%d,\"spin\":%d,", (failed || timed_out) ? "false" : "true", timed_out, g_nclients, g_nthreads, g_spin);
    printf("\"planned_input_lines\":%llu,\"input_lines\":%llu,\"acks\":%llu,", (unsigned long long)planned, (unsigned long long)sent, (unsigned long long)acks);
    printf("\"lines_total\":%llu,\"expected_lines\":%llu,\"deliveries\":%llu,\"bytes_total\":%llu,",
           (unsigned long long)lines, (unsigned long long)g_expect_lines,
           (unsigned long long)(lines >= acks ? lines - acks : 0), (unsigned long long)bytes);
    printf("\"wall_s\":%.6f,\"login_s\":%.3f,\"warm_s\":%.3f,", wall, (double)(t_login1 - t_login0) / 1e9, warm_s);
    printf("\"ack_latency_us\":{\"mean\":%.2f,\"p50\":%.2f,\"p99\":%.2f,\"max\":%.2f},",
           lat_mean / 1e3, nlat ? (double)lat[nlat / 2] / 1e3 : 0.0,
           nlat ? (double)lat[(size_t)((double)(nlat - 1) * 0.99)] / 1e3 : 0.0, nlat ? (double)lat[nlat - 1] / 1e3 : 0.0);
    /* acknowledgements that took more than ten medians (and at least 1 ms over it): how many, and how much wall
       clock they hold between them -- in a one-sender closed loop that is the stalled time itself */
    {
        double p50 = nlat ? (double)lat[nlat / 2] : 0.0, thr = p50 * 10.0 > p50 + 1e6 ? p50 * 10.0 : p50 + 1e6, tot = 0; size_t cnt = 0;
        for (size_t i = nlat; i-- > 0 && (double)lat[i] > thr;) { cnt++; tot += (double)lat[i]; }
        printf("\"slow_acks\":{\"threshold_us\":%.1f,\"count\":%zu,\"total_s\":%.6f},", thr / 1e3, cnt, tot / 1e9);
    }
    printf("\"progress_gaps\":{\"threshold_ms\":%.1f,\"count\":%u,\"total_s\":%.6f,\"max_ms\":%.3f},",
           (double)GAP_NS / 1e6, gap_n, (double)gap_total / 1e9, (double)gap_max / 1e6);
    printf("\"workers\":[");
    for (int t = 0; t < g_nthreads; t++) {
        struct worker *w = &g_workers[t]; int senders = 0;
        for (int i = 0; i < w->ncl; i++) if (w->cl[i]->timed.n) senders++;
        printf("%s{\"cpu\":%d,\"clients\":%d,\"senders\":%d,\"cpu_s\":%.6f,\"run_delay_s\":%.6f,\"window_s\":%.6f}", t ? "," : "", w->cpu, w->ncl, senders,
               w->cpu1_ns > w->cpu0_ns ? (double)(w->cpu1_ns - w->cpu0_ns) / 1e9 : 0.0,
               w->rq1_ns > w->rq0_ns ? (double)(w->rq1_ns - w->rq0_ns) / 1e9 : 0.0,
               w->w1_ns > w->w0_ns ? (double)(w->w1_ns - w->w0_ns) / 1e9 : 0.0);
    }
    printf("],");
    printf("\"servers\":[");
    for (int i = 0; i < g_nservers; i++) {
        printf("%s{\"pid\":%d,\"utime_s\":%.3f,\"stime_s\":%.3f,\"cpu_ns\":%llu,\"run_delay_ns\":%llu,"
               "\"voluntary_switches\":%llu,\"involuntary_switches\":%llu,"
               "\"read_syscalls\":%llu,\"write_syscalls\":%llu,\"bytes_written\":%llu}",
               i ? "," : "", g_server_pids[i],
               failed ? 0.0 : s1[i].utime_s - s0[i].utime_s, failed ? 0.0 : s1[i].stime_s - s0[i].stime_s,
               failed ? 0ull : (unsigned long long)(s1[i].sched_ns - s0[i].sched_ns),
               (unsigned long long)(s1[i].rq_ns - s0[i].rq_ns),
               (unsigned long long)(s1[i].vcsw - s0[i].vcsw), (unsigned long long)(s1[i].ivcsw - s0[i].ivcsw),
               (unsigned long long)(s1[i].syscr - s0[i].syscr), (unsigned long long)(s1[i].syscw - s0[i].syscw),
               (unsigned long long)(s1[i].wchar - s0[i].wchar));
    }
    printf("],\"per_client_lines\":[");
    for (int i = 0; i < g_nclients; i++) printf("%s%llu", i ? "," : "", (unsigned long long)g_clients[i].rx_lines);
    printf("]}\n");
    return (failed || timed_out) ? 1 : 0;
}
