/*
 * talker_port.c -- a minimal talker that restates the NUTS 3.3.3 input -> broadcast path.
 *
 * TEST INFRASTRUCTURE / ORACLE.  Built by oracle/Makefile into oracle/_build/talker_port.
 * It exists so that (1) the restated functions in nuts_path.c can be checked end to end,
 * byte for byte, against transcripts captured from the real reference build
 * (tests/golden/ *.json), and (2) bench.py has a CPU baseline of kind "port" on machines
 * where the reference binary is not available.  It is NOT a product and not a drop-in.
 *
 * Scope: exactly the rows of SURVEY.md section 8(a) plus what a client needs to get
 * there -- accept, the 3-stage login, look, go, the speech commands (say shout tell
 * emote semote pemote echo bcast wizshout), everything that changes fan-out results
 * (colour ignall ignshout igntell vis invis prompt mode afk, clones and their commands),
 * review/revtell, version, cls, and the NUTS netlink verbs config #5 exercises.  Every
 * other command name is recognised (so level gating behaves) and answered with a notice.
 * Boards, mail, ban administration, editor, pager state, timers and admin commands are out
 * of scope.
 *
 * Same algorithmic shape as the reference where the path is concerned: one select() loop,
 * one read() per ready socket per wake-up, first-line-only framing, per-recipient
 * transduction and one write(2) per recipient (two with colour on).  Own data layout:
 * users live in an ordered array, rooms in an array addressed by index.
 *
 * How it was written, stated plainly: the command handlers (say, shout, tell, emote, ... and the nl_* verbs in
 * talker_port_netlink.inc) are RESTATED FUNCTION BY FUNCTION FROM THE CITED LINES of nuts333.c -- the same checks
 * in the same order producing the same message texts, re-typed in C99 over arrays, often under the reference's own
 * names for the state they share (text, word_count, no_prompt, destructed, com_num, force_listen).  That closeness is
 * the point of an oracle and is what the byte-exact fixtures demand; it is not an independent design.  The event
 * loop, the data layout, the login FSM plumbing, the fast mode and the descriptor-limit handling are our own.
 */
#define _GNU_SOURCE
#include <arpa/inet.h>
#include <crypt.h>
#include <ctype.h>
#include <errno.h>
#include <fcntl.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <signal.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/select.h>
#include <sys/socket.h>
#include <sys/types.h>
#include <time.h>
#include <unistd.h>

#include "nuts_path.h"

#define VERSION "3.3.3"
#define MAX_ROOMS 32
#define MAX_LINKS 10
#define MAX_NETLINKS 16

enum { ACC_PUBLIC = 0, ACC_PRIVATE = 1, ACC_FIXED = 2, ACC_FIXED_PUBLIC = 2, ACC_FIXED_PRIVATE = 3 };  /* nuts333.h:45-49 */
enum { T_LOCAL = 0, T_CLONE = 1, T_REMOTE = 2 };                                                                   /* nuts333.h:57-59 */
enum { NL_UNCONNECTED, NL_INCOMING, NL_OUTGOING };                                                      /* nuts333.h:112-114 */
enum { ST_DOWN, ST_VERIFYING, ST_UP };
enum { ALLOW_ALL, ALLOW_IN, ALLOW_OUT };

struct netlink;

struct room {
    char name[21], label[6], desc[811], topic[61];
    char rev[NP_REVIEW_LINES][NP_REVIEW_LEN + 2];
    int revline, access, inlink, mesg_cnt;
    char link_label[MAX_LINKS][6];
    int link[MAX_LINKS], nlinks;
    char netlink_name[81];
    struct netlink *nl;
};

struct user {
    char name[13], desc[31], pass[26], in_phrase[41], out_phrase[41];
    char site[81], last_site[81];
    char buff[NP_ARR_SIZE], inpstr_old[NP_REVIEW_LEN + 1];
    char rev[NP_REVTELL_LINES][NP_REVIEW_LEN + 2];
    int revline;
    int room;                 /* index, -1 == away over a netlink (reference: room==NULL) */
    int invite_room;
    int type, port, site_port, login, sock, attempts, buffpos;
    int vis, ignall, ignshout, igntell, prompt, command_mode, muzzled, charmode_echo, colour, level;
    int remote_com;
    struct user *owner;       /* clones: the user whose socket they report to (nuts333.c:7143-7147) */
    int clone_hear;           /* 0 nothing, 1 swearing only, 2 everything (nuts333.h:60-62) */
    int afk;                  /* 0, 1 = any input resets, 2 = locked until the password is typed */
    char afk_mesg[61];
    time_t last_input, last_login, total_login, read_mail;
    int last_login_len;
    struct netlink *netlink, *pot_netlink;
};

struct netlink {
    char service[81], site[81], verification[21];
    char buffer[NP_ARR_SIZE * 2];
    int port, sock, type, stage, allow, lastcom;
    int ver_major, ver_minor, ver_patch;
    struct user *mesg_user;   /* (struct user*)-1: swallow until EMSG */
    int connect_room;
    int in_use;
};

/* ------------------------------------------------------------------ globals */
static struct room rooms[MAX_ROOMS]; static int nrooms;
static struct user **users; static int nusers, capusers;
static struct netlink netlinks[MAX_NETLINKS]; static int nnetlinks;

static char text[NP_TEXT_SIZE];
static char word[NP_MAX_WORDS][NP_WORD_LEN + 1];
static int word_count, com_num, force_listen, no_prompt, destructed;

static int port[3], listen_sock[3];
static char verification[81], confile[64] = "config";
static int max_users = 50, max_clones = 1, num_of_users, num_of_logins;
static int ban_swearing, colour_def = 1, prompt_def, charecho_def, allow_caps_in_name = 1;
static int system_logging = 1, password_echo, auto_connect = 1, min_private_users = 2;
static int gatecrash_level = NP_GOD + 1, wizport_level = NP_WIZ, minlogin_level = -1;
static int rem_user_maxlevel = NP_USER, rem_user_deflevel = NP_USER;
static int thour, tmin;

/* NUTS_PORT_FAST=1: the three CPU-side changes INTEGRATION.md section 3 proposes, so that their effect
 * can be MEASURED instead of estimated.  Same bytes on every socket (the parity tests run in both
 * modes); different work: (1) a broadcast is transduced once per colour variant, not once per
 * recipient; (2) the trailing colour reset travels in the same write(2) as the line; (3) netlink
 * sockets get TCP_NODELAY.  Off by default: the default build restates the reference's cost model. */
static int fast_mode;

static const char *level_name[] = { "NEW", "USER", "WIZ", "ARCH", "GOD" };
static const char *invisname = "A presence";

static void write_user(struct user *u, const char *str);
static void write_room_except(int rm, const char *str, struct user *except);
static void prompt(struct user *u);
static void look(struct user *u);
static void exec_com(struct user *u, char *inpstr);
static void disconnect_user(struct user *u);

/* ------------------------------------------------------------------ small helpers */
static void write_sock(int sock, const char *s) { if (write(sock, s, strlen(s)) < 0) { /* ignored, like nuts333.c:1285 */ } }

static void write_syslog(const char *str, int stamp)
{
    if (!system_logging) return;
    FILE *fp = fopen("syslog", "a");
    if (!fp) return;
    if (stamp) {
        time_t t = time(NULL); struct tm *tm = localtime(&t);
        fprintf(fp, "%02d/%02d %02d:%02d:%02d: %s", tm->tm_mday, tm->tm_mon + 1, tm->tm_hour, tm->tm_min, tm->tm_sec, str);
    } else fputs(str, fp);
    fclose(fp);
}

static void boot_exit(const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt);
    fprintf(stderr, "NUTS(port): "); vfprintf(stderr, fmt, ap); fputc('\n', stderr);
    va_end(ap);
    exit(1);
}

static int get_level(const char *name)
{
    for (int i = 0; i < 5; i++) if (!strcmp(level_name[i], name)) return i;
    return -1;
}

static void strtolower(char *s) { for (; *s; s++) *s = (char)tolower((unsigned char)*s); }

/* nuts333.c:2383-2391: first room whose name starts with the given text */
static int get_room(const char *name)
{
    size_t l = strlen(name);
    for (int i = 0; i < nrooms; i++) if (!strncmp(rooms[i].name, name, l)) return i;
    return -1;
}

/* nuts333.c:2362-2379: exact match first, then substring; capitalises the caller's buffer */
static struct user *get_user(char *name)
{
    name[0] = (char)toupper((unsigned char)name[0]);
    for (int i = 0; i < nusers; i++) { struct user *u = users[i]; if (!u->login && u->type != T_CLONE && !strcmp(u->name, name)) return u; }
    for (int i = 0; i < nusers; i++) { struct user *u = users[i]; if (!u->login && u->type != T_CLONE && strstr(u->name, name)) return u; }
    return NULL;
}

/* ------------------------------------------------------------------ output (L1) */
static void emit_to_sock(void *ctx, const char *buf, size_t len)
{
    if (write(*(int *)ctx, buf, len) < 0) { /* results ignored, nuts333.c:1318-1365 */ }
}

/* nuts333.c:1291-1366 */
static void write_user(struct user *u, const char *str)
{
    if (!u) return;
    if (u->type == T_REMOTE) {
        /* one MSG..EMSG frame per call, nuts333.c:1299-1306 */
        char stripped[NP_ARR_SIZE], mesg[NP_TEXT_SIZE + 64];
        if (u->netlink->ver_major <= 3 && u->netlink->ver_minor < 2) {
            np_colour_com_strip(str, stripped, sizeof(stripped)); str = stripped;
        }
        size_t l = strlen(str);
        if (!l || str[l - 1] != '\n') snprintf(mesg, sizeof(mesg), "MSG %s\n%s\nEMSG\n", u->name, str);
        else snprintf(mesg, sizeof(mesg), "MSG %s\n%sEMSG\n", u->name, str);
        write_sock(u->netlink->sock, mesg);
        return;
    }
    int sock = u->sock;
    if (fast_mode) {
        char out[NP_TEXT_SIZE * 6 + 64];
        size_t n = np_transduce(str, u->colour, out, sizeof(out));
        if (n && n <= sizeof(out)) { if (write(sock, out, n) < 0) {} return; }
    }
    np_write_user_stream(str, u->colour, emit_to_sock, &sock);
}

/* nuts333.c:1372-1385 */
static void write_level(int level, int above, const char *str, struct user *except)
{
    for (int i = 0; i < nusers; i++) {
        struct user *u = users[i];
        if (u == except || u->login || u->type == T_CLONE) continue;
        if ((above && u->level >= level) || (!above && u->level <= level)) write_user(u, str);
    }
}

/* the clone branch of the fan-out, nuts333.c:1416-1426: a clone relays what is said in ITS room to its
 * owner, prefixed with the room name; messages for every room (shouts, system lines) are skipped
 * because the owner hears those anyway */
static void clone_relay(struct user *u, int rm, const char *str)
{
    if (u->clone_hear == 0 || u->owner->ignall) return;
    if (rm != u->room) return;
    if (u->clone_hear == 1 && !np_contains_swearing(str)) return;
    char text2[NP_TEXT_SIZE + 64];
    snprintf(text2, sizeof(text2), "~FT[ %s ]:~RS %s", rooms[u->room].name, str);
    write_user(u->owner, text2);
}

/* nuts333.c:1401-1429 */
static void write_room_except(int rm, const char *str, struct user *except)
{
    if (fast_mode) {
        /* the transducer's output depends only on the recipient's colour bit: two variants, built lazily */
        static char variant[2][NP_TEXT_SIZE * 6 + 64];
        size_t len[2] = { 0, 0 }; int have[2] = { 0, 0 };
        for (int i = 0; i < nusers; i++) {
            struct user *u = users[i];
            struct np_listener l = { u->login, u->room >= 0, u->room == rm, u->ignall, u->ignshout, u == except };
            if (!np_fanout_admits(&l, rm < 0, force_listen, com_num)) continue;
            if (u->type == T_CLONE) { clone_relay(u, rm, str); continue; }
            if (u->type == T_REMOTE) { write_user(u, str); continue; }
            int c = u->colour ? 1 : 0;
            if (!have[c]) { len[c] = np_transduce(str, c, variant[c], sizeof(variant[c])); have[c] = 1; }
            if (len[c] && len[c] <= sizeof(variant[c])) { if (write(u->sock, variant[c], len[c]) < 0) {} }
            else write_user(u, str);
        }
        return;
    }
    for (int i = 0; i < nusers; i++) {
        struct user *u = users[i];
        struct np_listener l = { u->login, u->room >= 0, u->room == rm, u->ignall, u->ignshout, u == except };
        if (!np_fanout_admits(&l, rm < 0, force_listen, com_num)) continue;
        if (u->type == T_CLONE) clone_relay(u, rm, str);
        else write_user(u, str);
    }
}
static void write_room(int rm, const char *str) { write_room_except(rm, str, NULL); }

static void more_emit(void *ctx, const char *buf, size_t len) { if (write(*(int *)ctx, buf, len) < 0) {} }

/* nuts333.c:2205-2322, the two uses on the login path: motd1 before login (user NULL: no colour, no paging) and
 * motd2 after it.  A file longer than a screen would enter the pager state, which is out of scope (DESIGN.md
 * section 8): the whole file is sent.  Every line of the file goes through ONE staging buffer that
 * is flushed when full, before a newline or '~' once fewer than 6 bytes remain, and when the file ends -- the
 * same write(2) boundaries as the reference, not one write per line; no end-of-string colour reset. */
static int more(struct user *u, int sock, const char *filename)
{
    FILE *fp = fopen(filename, "r");
    if (!fp) return 0;
    char line[NP_TEXT_SIZE];
    int colour = u ? u->colour : 0;
    struct np_stage st; np_stage_init(&st);
    while (fgets(line, sizeof(line) - 1, fp)) {
        if (feof(fp) && line[strlen(line) - 1] != '\n') break;   /* unterminated last line is dropped (c:2236) */
        np_stage_feed(&st, line, colour, more_emit, &sock);
    }
    np_stage_flush(&st, more_emit, &sock);
    fclose(fp);
    return 2;
}

/* ------------------------------------------------------------------ objects (L0) */
static struct user *create_user(void)
{
    struct user *u = calloc(1, sizeof(*u));
    if (!u) return NULL;
    if (nusers == capusers) {
        capusers = capusers ? capusers * 2 : 64;
        users = realloc(users, sizeof(*users) * (size_t)capusers);
    }
    users[nusers++] = u;
    /* defaults of nuts333.c:2693-2747 */
    u->type = T_LOCAL; u->room = -1; u->invite_room = -1; u->sock = -1; u->vis = 1; u->remote_com = -1;
    u->read_mail = u->last_input = u->last_login = time(NULL);
    u->prompt = prompt_def; u->colour = colour_def; u->charmode_echo = charecho_def;
    u->clone_hear = 2;
    return u;
}

static void destruct_user(struct user *u)
{
    for (int i = 0; i < nusers; i++) {
        if (users[i] == u) {
            memmove(&users[i], &users[i + 1], sizeof(*users) * (size_t)(nusers - i - 1));
            nusers--;
            break;
        }
    }
    for (int i = 0; i < nnetlinks; i++) if (netlinks[i].mesg_user == u) netlinks[i].mesg_user = (struct user *)-1;
    free(u);
    destructed = 1;
}

static int user_index(const struct user *u)
{
    for (int i = 0; i < nusers; i++) if (users[i] == u) return i;
    return -1;
}

/* nuts333.c:2870-2882 */
static void destroy_user_clones(struct user *owner)
{
    for (int i = 0; i < nusers;) {
        struct user *u = users[i];
        if (u->type == T_CLONE && u->owner == owner) {
            snprintf(text, sizeof(text), "The clone of %s shimmers and vanishes.\n", u->name);
            write_room_except(u->room, text, NULL);
            destruct_user(u);
            continue;
        }
        i++;
    }
}

/* nuts333.c:2087-2107 */
static void clear_revbuff(struct room *r) { for (int i = 0; i < NP_REVIEW_LINES; i++) r->rev[i][0] = 0; r->revline = 0; }

static void reset_access(int rm)
{
    if (rm < 0 || rooms[rm].access != ACC_PRIVATE) return;
    int cnt = 0;
    for (int i = 0; i < nusers; i++) if (users[i]->room == rm) cnt++;
    if (cnt < min_private_users) {
        write_room(rm, "Room access returned to ~FGPUBLIC.\n");
        rooms[rm].access = ACC_PUBLIC;
        for (int i = 0; i < nusers; i++) if (users[i]->invite_room == rm) users[i]->invite_room = -1;
        clear_revbuff(&rooms[rm]);
    }
}

/* nuts333.c:2412-2421 */
static int has_room_access(const struct user *u, int rm)
{
    const struct room *r = &rooms[rm];
    if ((r->access & ACC_PRIVATE) && u->level < gatecrash_level && u->invite_room != rm
        && !((r->access & ACC_FIXED) && u->level >= NP_WIZ)) return 0;
    return 1;
}

/* ------------------------------------------------------------------ user records */
/* nuts333.c:1611-1640; format DOCS/userdata_format:7-15 */
static int load_user_details(struct user *u)
{
    char fn[128], line[128];
    snprintf(fn, sizeof(fn), "userfiles/%s.D", u->name);
    FILE *fp = fopen(fn, "r");
    if (!fp) return 0;
    int t1 = 0, t2 = 0, t3 = 0;
    if (fscanf(fp, "%25s", u->pass) != 1) { fclose(fp); return 0; }
    if (fscanf(fp, "%d %d %d %d %d %d %d %d %d %d", &t1, &t2, &u->last_login_len, &t3, &u->level, &u->prompt,
               &u->muzzled, &u->charmode_echo, &u->command_mode, &u->colour) != 10) { fclose(fp); return 0; }
    u->last_login = t1; u->total_login = t2; u->read_mail = t3;
    if (fscanf(fp, "%80s\n", u->last_site) != 1) u->last_site[0] = 0;
    struct { char *dst; int n; } f[3] = { { u->desc, 30 }, { u->in_phrase, 40 }, { u->out_phrase, 40 } };
    for (int i = 0; i < 3; i++) {
        line[0] = 0;
        if (fgets(line, f[i].n + 2, fp)) { size_t l = strlen(line); if (l) line[l - 1] = 0; }
        strcpy(f[i].dst, line);
    }
    fclose(fp);
    return 1;
}

/* nuts333.c:1645-1673 */
static int save_user_details(struct user *u, int save_current)
{
    if (u->type == T_REMOTE) return 0;
    char fn[128]; snprintf(fn, sizeof(fn), "userfiles/%s.D", u->name);
    FILE *fp = fopen(fn, "w");
    if (!fp) return 0;
    fprintf(fp, "%s\n", u->pass);
    if (save_current) fprintf(fp, "%d %d %d ", (int)time(NULL), (int)u->total_login, (int)(time(NULL) - u->last_login));
    else fprintf(fp, "%d %d %d ", (int)u->last_login, (int)u->total_login, u->last_login_len);
    fprintf(fp, "%d %d %d %d %d %d %d\n", (int)u->read_mail, u->level, u->prompt, u->muzzled, u->charmode_echo, u->command_mode, u->colour);
    fprintf(fp, "%s\n%s\n%s\n%s\n", save_current ? u->site : u->last_site, u->desc, u->in_phrase, u->out_phrase);
    fclose(fp);
    return 1;
}

static int listed_in(const char *file, const char *needle, int substring)
{
    char fn[128], line[128]; snprintf(fn, sizeof(fn), "datafiles/%s", file);
    FILE *fp = fopen(fn, "r");
    if (!fp) return 0;
    int hit = 0;
    while (!hit && fscanf(fp, "%127s", line) == 1) hit = substring ? strstr(needle, line) != NULL : !strcmp(line, needle);
    fclose(fp);
    return hit;
}

static int has_unread_mail(const struct user *u)
{
    char fn[128]; snprintf(fn, sizeof(fn), "userfiles/%s.M", u->name);
    FILE *fp = fopen(fn, "r");
    if (!fp) return 0;
    int tm = 0; if (fscanf(fp, "%d", &tm) != 1) tm = 0;
    fclose(fp);
    return tm > (int)u->read_mail;
}

/* ------------------------------------------------------------------ login / logout */
/* nuts333.c:1814-1834 */
static void echo_off(struct user *u) { if (!password_echo) write_user(u, "\377\373\001"); }
static void echo_on(struct user *u) { if (!password_echo) write_user(u, "\377\374\001"); }

static void attempts(struct user *u)
{
    if (++u->attempts == 3) { write_user(u, "\nMaximum attempts reached.\n\n"); disconnect_user(u); return; }
    u->login = 3; u->pass[0] = 0;
    write_user(u, "Give me a name: ");
    echo_on(u);
}

/* nuts333.c:1677-1759 (session swap / remote pull-back branches out of scope) */
static void connect_user(struct user *u)
{
    char temp[40];
    snprintf(text, sizeof(text), "~OLSIGN ON:~RS %s %s\n", u->name, u->desc);
    write_level(NP_USER, 0, text, NULL);
    snprintf(text, sizeof(text), "~OLSIGN ON:~RS %s %s  ~RS~FT(%s:%d)\n", u->name, u->desc, u->site, u->site_port);
    write_level(NP_WIZ, 1, text, NULL);

    write_user(u, "\n");
    more(u, u->sock, "motd2");
    if (u->last_site[0]) {
        snprintf(temp, sizeof(temp), "%s", ctime(&u->last_login));
        temp[strlen(temp) - 1] = 0;
        snprintf(text, sizeof(text), "Welcome %s...\n\n~BBYou were last logged in on %s from %s.\n\n", u->name, temp, u->last_site);
    } else snprintf(text, sizeof(text), "Welcome %s...\n\n", u->name);
    write_user(u, text);
    u->room = 0;
    u->last_login = time(NULL);
    snprintf(text, sizeof(text), "~FTYour level is:~RS~OL %s\n", level_name[u->level]);
    write_user(u, text);
    look(u);
    if (has_unread_mail(u)) write_user(u, "\07~FT~OL~LI** YOU HAVE UNREAD MAIL **\n");
    prompt(u);
    snprintf(text, sizeof(text), "%s logged in on port %d from %s:%d.\n", u->name, u->port, u->site, u->site_port);
    write_syslog(text, 1);
    num_of_users++; num_of_logins--;
    u->login = 0;
}

/* nuts333.c:1451-1589 */
static void login(struct user *u, const char *inpstr)
{
    char name[NP_ARR_SIZE] = "", passwd[NP_ARR_SIZE] = "";
    switch (u->login) {
    case 3:
        sscanf(inpstr, "%999s", name);
        if ((signed char)name[0] < 33) { write_user(u, "\nGive me a name: "); return; }
        if (!strcmp(name, "quit")) { write_user(u, "\n\n*** Abandoning login attempt ***\n\n"); disconnect_user(u); return; }
        if (!strcmp(name, "who")) { write_user(u, "\n[talker_port: who is outside the restated path]\n\nGive me a name: "); return; }
        if (!strcmp(name, "version")) { snprintf(text, sizeof(text), "\nNUTS version %s\n\nGive me a name: ", VERSION); write_user(u, text); return; }
        if (strlen(name) < 3) { write_user(u, "\nName too short.\n\n"); attempts(u); return; }
        if (strlen(name) > 12) { write_user(u, "\nName too long.\n\n"); attempts(u); return; }
        for (size_t i = 0; name[i]; i++)
            if (!isalpha((unsigned char)name[i])) { write_user(u, "\nOnly letters are allowed in a name.\n\n"); attempts(u); return; }
        if (!allow_caps_in_name) strtolower(name);
        name[0] = (char)toupper((unsigned char)name[0]);
        if (listed_in("userban", name, 0)) {
            write_user(u, "\nYou are banned from this talker.\n\n"); disconnect_user(u); return;
        }
        strcpy(u->name, name);
        for (int i = 0; i < nusers; i++)
            if (users[i]->login && users[i] != u && !strcmp(users[i]->name, u->name)) { disconnect_user(users[i]); break; }
        if (!load_user_details(u)) {
            if (u->port == port[1]) { write_user(u, "\nSorry, new logins cannot be created on this port.\n\n"); disconnect_user(u); return; }
            if (minlogin_level > -1) { write_user(u, "\nSorry, new logins cannot be created at this time.\n\n"); disconnect_user(u); return; }
            write_user(u, "New user...\n");
        } else {
            if (u->port == port[1] && u->level < wizport_level) {
                snprintf(text, sizeof(text), "\nSorry, only users of level %s and above can log in on this port.\n\n", level_name[wizport_level]);
                write_user(u, text); disconnect_user(u); return;
            }
            if (u->level < minlogin_level) { write_user(u, "\nSorry, the talker is locked out to users of your level.\n\n"); disconnect_user(u); return; }
        }
        write_user(u, "Give me a password: ");
        echo_off(u);
        u->login = 2;
        return;
    case 2:
        sscanf(inpstr, "%999s", passwd);
        if (strlen(passwd) < 3) { write_user(u, "\n\nPassword too short.\n\n"); attempts(u); return; }
        if (strlen(passwd) > 20) { write_user(u, "\n\nPassword too long.\n\n"); attempts(u); return; }
        if (!u->pass[0]) {
            snprintf(u->pass, sizeof(u->pass), "%s", crypt(passwd, "NU"));
            write_user(u, "\nPlease confirm password: ");
            u->login = 1;
        } else {
            const char *h = crypt(passwd, "NU");
            if (h && !strcmp(u->pass, h)) { echo_on(u); connect_user(u); return; }
            write_user(u, "\n\nIncorrect login.\n\n");
            attempts(u);
        }
        return;
    case 1: {
        sscanf(inpstr, "%999s", passwd);
        const char *h = crypt(passwd, "NU");
        if (!h || strcmp(u->pass, h)) { write_user(u, "\n\nPasswords do not match.\n\n"); attempts(u); return; }
        echo_on(u);
        strcpy(u->desc, "hasn't used .desc yet"); strcpy(u->in_phrase, "enters"); strcpy(u->out_phrase, "goes");
        u->last_site[0] = 0; u->level = 0; u->muzzled = 0; u->command_mode = 0;
        u->prompt = prompt_def; u->colour = colour_def; u->charmode_echo = charecho_def;
        save_user_details(u, 1);
        snprintf(text, sizeof(text), "New user \"%s\" created.\n", u->name);
        write_syslog(text, 1);
        connect_user(u);
    } }
}

/* nuts333.c:1763-1810.  (The reference clears `destructed` again at the end, which is what
 * makes its .quit read freed memory; we keep the flag set.) */
static void disconnect_user(struct user *u)
{
    int rm = u->room;
    if (u->login) { close(u->sock); destruct_user(u); num_of_logins--; return; }
    if (u->type != T_REMOTE) {
        save_user_details(u, 1);
        snprintf(text, sizeof(text), "%s logged out.\n", u->name); write_syslog(text, 1);
        write_user(u, "\n~OL~FBYou are removed from this reality...\n\n");
        close(u->sock); u->sock = -1;
        snprintf(text, sizeof(text), "~OLSIGN OFF:~RS %s %s\n", u->name, u->desc);
        write_room_except(-1, text, u);
        if (u->room < 0 && u->netlink) { snprintf(text, sizeof(text), "REL %s\n", u->name); write_sock(u->netlink->sock, text); }
    } else {
        write_user(u, "\n~FR~OLYou are pulled back in disgrace to your own domain...\n");
        snprintf(text, sizeof(text), "REMVD %s\n", u->name); write_sock(u->netlink->sock, text);
        snprintf(text, sizeof(text), "~FR~OL%s is banished from here!\n", u->name);
        write_room_except(rm, text, u);
    }
    num_of_users--;
    destroy_user_clones(u);
    destruct_user(u);
    reset_access(rm);
}

/* ------------------------------------------------------------------ prompt / look / go */
/* nuts333.c:2174-2197 */
static void prompt(struct user *u)
{
    if (no_prompt) return;
    if (u->type == T_REMOTE) { snprintf(text, sizeof(text), "PRM %s\n", u->name); write_sock(u->netlink->sock, text); return; }
    if (u->command_mode) { write_user(u, u->vis ? "~FTCOM> " : "~FTCOM+> "); return; }
    if (!u->prompt) return;
    int el = (int)(time(NULL) - u->last_login);
    snprintf(text, sizeof(text), "~FT<%02d:%02d, %02d:%02d, %s%s>\n", thour, tmin, el / 3600, (el % 3600) / 60, u->name, u->vis ? "" : "+");
    write_user(u, text);
}

/* nuts333.c:3942-4004 */
static void look(struct user *u)
{
    struct room *rm = &rooms[u->room];
    char temp[128];
    snprintf(text, sizeof(text), "\n~FTRoom: %s%s\n\n", (rm->access & ACC_PRIVATE) ? "~FR" : "~FG", rm->name);
    write_user(u, text);
    write_user(u, rm->desc);
    strcpy(text, "\n~FTExits are:");
    for (int i = 0; i < rm->nlinks; i++) {
        struct room *l = &rooms[rm->link[i]];
        snprintf(temp, sizeof(temp), "  %s%s", (l->access & ACC_PRIVATE) ? "~FR" : "~FG", l->name);
        strcat(text, temp);
    }
    if (rm->nl && rm->nl->stage == ST_UP) {
        snprintf(temp, sizeof(temp), "  %s%s*", rm->nl->allow == ALLOW_IN ? "~FR" : "~FG", rm->nl->service);
        strcat(text, temp);
    } else if (!rm->nlinks) strcpy(text, "\n~FTThere are no exits.");
    strcat(text, "\n\n");
    write_user(u, text);

    int seen = 0;
    for (int i = 0; i < nusers; i++) {
        struct user *o = users[i];
        if (o->room != u->room || o == u || (!o->vis && o->level > u->level)) continue;
        if (!seen++) write_user(u, "~FTYou can see:\n");
        const char *afk = o->afk ? "~BR(AFK)" : "";
        if (!o->vis) snprintf(text, sizeof(text), "     ~FR*~RS%s %s~RS  %s\n", o->name, o->desc, afk);
        else snprintf(text, sizeof(text), "      %s %s~RS  %s\n", o->name, o->desc, afk);
        write_user(u, text);
    }
    if (!seen) write_user(u, "~FTYou are all alone here.\n");
    write_user(u, "\n");

    strcpy(text, "Access is ");
    switch (rm->access) {
    case ACC_PUBLIC: strcat(text, "set to ~FGPUBLIC~RS"); break;
    case ACC_PRIVATE: strcat(text, "set to ~FRPRIVATE~RS"); break;
    case ACC_FIXED_PUBLIC: strcat(text, "~FRfixed~RS to ~FGPUBLIC~RS"); break;
    case ACC_FIXED_PRIVATE: strcat(text, "~FRfixed~RS to ~FRPRIVATE~RS"); break;
    }
    snprintf(temp, sizeof(temp), " and there are ~OL~FM%d~RS messages on the board.\n", rm->mesg_cnt);
    strcat(text, temp);
    write_user(u, text);
    if (rm->topic[0]) { snprintf(text, sizeof(text), "Current topic: %s\n", rm->topic); write_user(u, text); return; }
    write_user(u, "No topic has been set yet.\n");
}

/* nuts333.c:4409-4459 (teleport==2, the .move victim case, is out of scope) */
static void move_user(struct user *u, int rm, int teleport)
{
    int old = u->room;
    if (!has_room_access(u, rm)) { write_user(u, "That room is currently private, you cannot enter.\n"); return; }
    if (u->invite_room == rm) u->invite_room = -1;
    if (!u->vis) {
        write_room(rm, "A presence enters the room...\n");
        write_room_except(old, "A presence leaves the room.\n", u);
    } else if (teleport) {
        snprintf(text, sizeof(text), "~FT~OL%s appears in an explosion of blue magic!\n", u->name); write_room(rm, text);
        snprintf(text, sizeof(text), "~FT~OL%s chants a spell and vanishes into a magical blue vortex!\n", u->name);
        write_room_except(old, text, u);
    } else {
        snprintf(text, sizeof(text), "%s %s.\n", u->name, u->in_phrase); write_room(rm, text);
        snprintf(text, sizeof(text), "%s %s to the %s.\n", u->name, u->out_phrase, rooms[rm].name);
        write_room_except(old, text, u);
    }
    u->room = rm;
    look(u);
    reset_access(old);
}

/* nuts333.c:4305-4405 */
static void go(struct user *u)
{
    if (word_count < 2) { write_user(u, "Go where?\n"); return; }
    struct netlink *nl = rooms[u->room].nl;
    if (nl && !strncmp(nl->service, word[1], strlen(word[1]))) {
        if (u->pot_netlink == nl) { write_user(u, "The remote service may be lagged, please be patient...\n"); return; }
        int rm = u->room;
        if (nl->stage < ST_UP) { write_user(u, "The netlink is inactive.\n"); return; }
        if (nl->allow == ALLOW_IN && u->netlink != nl) { write_user(u, "Sorry, link is for incoming users only.\n"); return; }
        if (u->netlink == nl) {
            /* a remote user going home: tell the home site we removed him */
            write_user(u, "~FB~OLYou traverse cyberspace...\n");
            snprintf(text, sizeof(text), "REMVD %s\n", u->name); write_sock(nl->sock, text);
            if (u->vis) { snprintf(text, sizeof(text), "%s goes to the %s\n", u->name, nl->service); write_room_except(rm, text, u); }
            else write_room_except(rm, "A presence leaves the room.\n", u);
            destroy_user_clones(u);
            destruct_user(u); reset_access(rm); num_of_users--; no_prompt = 1;
            return;
        }
        if (u->type == T_REMOTE) { write_user(u, "Sorry, due to software limitations you can only traverse one netlink.\n"); return; }
        const char *pw = word[2][0] ? crypt(word[2], "NU") : u->pass;
        if (nl->ver_major <= 3 && nl->ver_minor <= 3 && nl->ver_patch < 1)
            snprintf(text, sizeof(text), "TRANS %s %s %s\n", u->name, pw, u->desc);
        else snprintf(text, sizeof(text), "TRANS %s %s %d %s\n", u->name, pw, u->level, u->desc);
        write_sock(nl->sock, text);
        u->remote_com = NP_GO; u->pot_netlink = nl; no_prompt = 1;
        return;
    }
    if (u->remote_com == NP_GO) {
        snprintf(text, sizeof(text), "REL %s\n", u->name); write_sock(u->pot_netlink->sock, text);
        u->remote_com = -1; u->pot_netlink = NULL;
    }
    int rm = get_room(word[1]);
    if (rm < 0) { write_user(u, "There is no such room.\n"); return; }
    if (rm == u->room) { snprintf(text, sizeof(text), "You are already in the %s!\n", rooms[rm].name); write_user(u, text); return; }
    for (int i = 0; i < rooms[u->room].nlinks; i++) if (rooms[u->room].link[i] == rm) { move_user(u, rm, 0); return; }
    if (u->level < NP_WIZ) { snprintf(text, sizeof(text), "The %s is not adjoined to here.\n", rooms[rm].name); write_user(u, text); return; }
    move_user(u, rm, 1);
}

/* ------------------------------------------------------------------ speech (L2) */
static const char *shown_name(const struct user *u) { return u->vis ? u->name : invisname; }

/* nuts333.c:4062-4100 */
static void say(struct user *u, const char *inpstr)
{
    if (u->muzzled) { write_user(u, "You are muzzled, you cannot speak.\n"); return; }
    if (u->room < 0) {
        snprintf(text, sizeof(text), "ACT %s say %s\n", u->name, inpstr); write_sock(u->netlink->sock, text);
        no_prompt = 1; return;
    }
    if (word_count < 2 && u->command_mode) { write_user(u, "Say what?\n"); return; }
    const char *verb = np_say_verb(inpstr);
    if (u->type == T_CLONE) {                                     /* c:4085-4090: no swear check, no "You say" */
        snprintf(text, sizeof(text), "Clone of %s %ss: %s\n", u->name, verb, inpstr);
        write_room(u->room, text);
        np_record(&rooms[u->room].rev[0][0], NP_REVIEW_LINES, &rooms[u->room].revline, text);
        return;
    }
    if (ban_swearing && np_contains_swearing(inpstr)) { write_user(u, "Swearing is not allowed here.\n"); return; }
    snprintf(text, sizeof(text), "You %s: %s\n", verb, inpstr);
    write_user(u, text);
    snprintf(text, sizeof(text), "%s %ss: %s\n", shown_name(u), verb, inpstr);
    write_room_except(u->room, text, u);
    np_record(&rooms[u->room].rev[0][0], NP_REVIEW_LINES, &rooms[u->room].revline, text);
}

/* nuts333.c:4104-4124 */
static void shout(struct user *u, const char *inpstr)
{
    if (u->muzzled) { write_user(u, "You are muzzled, you cannot shout.\n"); return; }
    if (word_count < 2) { write_user(u, "Shout what?\n"); return; }
    if (ban_swearing && np_contains_swearing(inpstr)) { write_user(u, "Swearing is not allowed here.\n"); return; }
    snprintf(text, sizeof(text), "~OLYou shout:~RS %s\n", inpstr);
    write_user(u, text);
    snprintf(text, sizeof(text), "~OL%s shouts:~RS %s\n", shown_name(u), inpstr);
    write_room_except(-1, text, u);
}

/* shared early-outs of tell and pemote, nuts333.c:4149-4172 / 4251-4273 */
static int private_blocked(struct user *u, struct user *t, const char *what)
{
    if (t->afk) {
        if (t->afk_mesg[0]) snprintf(text, sizeof(text), "%s is AFK, message is: %s\n", t->name, t->afk_mesg);
        else snprintf(text, sizeof(text), "%s is AFK at the moment.\n", t->name);
        write_user(u, text); return 1;
    }
    if (t->ignall && (u->level < NP_WIZ || t->level > u->level)) {
        snprintf(text, sizeof(text), "%s is ignoring everyone at the moment.\n", t->name); write_user(u, text); return 1;
    }
    if (t->igntell && (u->level < NP_WIZ || t->level > u->level)) {
        snprintf(text, sizeof(text), "%s is ignoring %s at the moment.\n", t->name, what); write_user(u, text); return 1;
    }
    if (t->room < 0) {
        snprintf(text, sizeof(text), "%s is offsite and would not be able to reply to you.\n", t->name); write_user(u, text); return 1;
    }
    return 0;
}

/* nuts333.c:4128-4182 */
static void tell(struct user *u, const char *inpstr)
{
    if (u->muzzled) { write_user(u, "You are muzzled, you cannot tell anyone anything.\n"); return; }
    if (word_count < 3) { write_user(u, "Tell who what?\n"); return; }
    struct user *t = get_user(word[1]);
    if (!t) { write_user(u, "There is no one of that name logged on.\n"); return; }
    if (t == u) { write_user(u, "Talking to yourself is the first sign of madness.\n"); return; }
    if (private_blocked(u, t, "tells")) return;
    inpstr = np_remove_first(inpstr);
    size_t l = strlen(inpstr);
    const char *verb = (l && inpstr[l - 1] == '?') ? "ask" : "tell";
    snprintf(text, sizeof(text), "~OLYou %s %s:~RS %s\n", verb, t->name, inpstr);
    write_user(u, text);
    snprintf(text, sizeof(text), "~OL%s %ss you:~RS %s\n", shown_name(u), verb, inpstr);
    write_user(t, text);
    np_record(&t->rev[0][0], NP_REVTELL_LINES, &t->revline, text);
}

/* nuts333.c:4186-4206 */
static void emote(struct user *u, const char *inpstr)
{
    if (u->muzzled) { write_user(u, "You are muzzled, you cannot emote.\n"); return; }
    if (word_count < 2 && (signed char)inpstr[1] < 33) { write_user(u, "Emote what?\n"); return; }
    if (ban_swearing && np_contains_swearing(inpstr)) { write_user(u, "Swearing is not allowed here.\n"); return; }
    if (inpstr[0] == ';') snprintf(text, sizeof(text), "%s%s\n", shown_name(u), inpstr + 1);
    else snprintf(text, sizeof(text), "%s %s\n", shown_name(u), inpstr);
    write_room(u->room, text);
    np_record(&rooms[u->room].rev[0][0], NP_REVIEW_LINES, &rooms[u->room].revline, text);
}

/* nuts333.c:4210-4226 */
static void semote(struct user *u, const char *inpstr)
{
    if (u->muzzled) { write_user(u, "You are muzzled, you cannot emote.\n"); return; }
    if (word_count < 2 && (signed char)inpstr[1] < 33) { write_user(u, "Shout emote what?\n"); return; }
    if (inpstr[0] == '#') snprintf(text, sizeof(text), "~OL!!~RS %s%s\n", shown_name(u), inpstr + 1);
    else snprintf(text, sizeof(text), "~OL!!~RS %s %s\n", shown_name(u), inpstr);
    write_room(-1, text);
}

/* nuts333.c:4230-4281 */
static void pemote(struct user *u, const char *inpstr)
{
    if (u->muzzled) { write_user(u, "You are muzzled, you cannot emote.\n"); return; }
    if (word_count < 3) { write_user(u, "Private emote what?\n"); return; }
    word[1][0] = (char)toupper((unsigned char)word[1][0]);
    if (!strcmp(word[1], u->name)) { write_user(u, "Emoting to yourself is the second sign of madness.\n"); return; }
    struct user *t = get_user(word[1]);
    if (!t) { write_user(u, "There is no one of that name logged on.\n"); return; }
    if (private_blocked(u, t, "private emotes")) return;
    inpstr = np_remove_first(inpstr);
    snprintf(text, sizeof(text), "~OL(To %s)~RS %s %s\n", t->name, shown_name(u), inpstr);
    write_user(u, text);
    snprintf(text, sizeof(text), "~OL>>~RS %s %s\n", shown_name(u), inpstr);
    write_user(t, text);
    np_record(&t->rev[0][0], NP_REVTELL_LINES, &t->revline, text);
}

/* nuts333.c:4285-4300 */
static void echo(struct user *u, const char *inpstr)
{
    if (u->muzzled) { write_user(u, "You are muzzled, you cannot echo.\n"); return; }
    if (word_count < 2) { write_user(u, "Echo what?\n"); return; }
    snprintf(text, sizeof(text), "(%s) ", u->name);
    write_level(NP_WIZ, 1, text, NULL);
    snprintf(text, sizeof(text), "- %s\n", inpstr);
    write_room(u->room, text);
    np_record(&rooms[u->room].rev[0][0], NP_REVIEW_LINES, &rooms[u->room].revline, text);
}

/* nuts333.c:5192-5223 */
static void review(struct user *u)
{
    int rm = u->room;
    if (word_count >= 2) {
        if ((rm = get_room(word[1])) < 0) { write_user(u, "There is no such room.\n"); return; }
        if (!has_room_access(u, rm)) { write_user(u, "That room is currently private, you cannot review the conversation.\n"); return; }
    }
    struct room *r = &rooms[rm];
    int cnt = 0;
    for (int i = 0; i < NP_REVIEW_LINES; i++) {
        int line = (r->revline + i) % NP_REVIEW_LINES;
        if (!r->rev[line][0]) continue;
        if (!cnt++) { snprintf(text, sizeof(text), "\n~BB~FG*** Review buffer for the %s ***\n\n", r->name); write_user(u, text); }
        write_user(u, r->rev[line]);
    }
    write_user(u, cnt ? "\n~BB~FG*** End ***\n\n" : "Review buffer is empty.\n");
}

/* nuts333.c:7699-7715 */
static void revtell(struct user *u)
{
    int cnt = 0;
    for (int i = 0; i < NP_REVTELL_LINES; i++) {
        int line = (u->revline + i) % NP_REVTELL_LINES;
        if (!u->rev[line][0]) continue;
        if (!cnt++) write_user(u, "\n~BB~FG*** Your revtell buffer ***\n\n");
        write_user(u, u->rev[line]);
    }
    write_user(u, cnt ? "\n~BB~FG*** End ***\n\n" : "Revtell buffer is empty.\n");
}

/* nuts333.c:4772-4788: heard even by users who ignore everything (force_listen) */
static void bcast(struct user *u, const char *inpstr)
{
    if (word_count < 2) { write_user(u, "Usage: bcast <message>\n"); return; }
    if (u->muzzled) { write_user(u, "You are muzzled, you cannot broadcast anything.\n"); return; }
    force_listen = 1;
    if (u->vis) snprintf(text, sizeof(text), "\07\n~BR*** Broadcast message from %s ***\n%s\n\n", u->name, inpstr);
    else snprintf(text, sizeof(text), "\07\n~BR*** Broadcast message ***\n%s\n\n", inpstr);
    write_room(-1, text);
}

/* nuts333.c:6527-6565 */
static void wizshout(struct user *u, const char *inpstr)
{
    if (u->muzzled) { write_user(u, "You are muzzled, you cannot wizshout.\n"); return; }
    if (word_count < 2) { write_user(u, "Usage: wizshout [<superuser level>] <message>\n"); return; }
    if (ban_swearing && np_contains_swearing(inpstr)) { write_user(u, "Swearing is not allowed here.\n"); return; }
    for (char *p = word[1]; *p; p++) *p = (char)toupper((unsigned char)*p);
    int lev = get_level(word[1]);
    if (lev == -1) {
        snprintf(text, sizeof(text), "~OLYou wizshout:~RS %s\n", inpstr); write_user(u, text);
        snprintf(text, sizeof(text), "~OL%s wizshouts:~RS %s\n", u->name, inpstr);
        write_level(NP_WIZ, 1, text, u);
        return;
    }
    if (lev < NP_WIZ || word_count < 3) { write_user(u, "Usage: wizshout [<superuser level>] <message>\n"); return; }
    if (lev > u->level) { write_user(u, "You cannot specifically shout to users of a higher level than yourself.\n"); return; }
    inpstr = np_remove_first(inpstr);
    snprintf(text, sizeof(text), "~OLYou wizshout to level %s:~RS %s\n", level_name[lev], inpstr); write_user(u, text);
    snprintf(text, sizeof(text), "~OL%s wizshouts to level %s:~RS %s\n", u->name, level_name[lev], inpstr);
    write_level(lev, 1, text, u);
}

/* nuts333.c:7409-7454 */
static void afk(struct user *u, const char *inpstr)
{
    if (word_count > 1) {
        int lock = !strcmp(word[1], "lock");
        if (lock) {
            if (u->type == T_REMOTE) { write_user(u, "Sorry, due to software limitations remote users cannot use the lock option.\n"); return; }
            inpstr = np_remove_first(inpstr);
        }
        if (strlen(inpstr) > 60) { write_user(u, "AFK message too long.\n"); return; }
        write_user(u, lock ? "You are now AFK with the session locked, enter your password to unlock it.\n"
                           : "You are now AFK, press <return> to reset.\n");
        if (inpstr[0]) { strcpy(u->afk_mesg, inpstr); write_user(u, "AFK message set.\n"); }
        u->afk = lock ? 2 : 1;
    } else {
        write_user(u, "You are now AFK, press <return> to reset.\n");
        u->afk = 1;
    }
    if (u->vis) {
        if (u->afk_mesg[0]) snprintf(text, sizeof(text), "%s goes AFK: %s\n", u->name, u->afk_mesg);
        else snprintf(text, sizeof(text), "%s goes AFK...\n", u->name);
        write_room_except(u->room, text, u);
    }
}

/* ---- clones (nuts333.c:7100-7357): listener objects that share their owner's socket ---- */
static struct user *find_clone(const struct user *owner, int rm)
{
    for (int i = 0; i < nusers; i++) if (users[i]->type == T_CLONE && users[i]->room == rm && users[i]->owner == owner) return users[i];
    return NULL;
}

/* nuts333.c:7100-7162 */
static void create_clone(struct user *u)
{
    int rm = u->room;
    if (word_count >= 2 && (rm = get_room(word[1])) < 0) { write_user(u, "There is no such room.\n"); return; }
    if (!has_room_access(u, rm)) { write_user(u, "That room is currently private, you cannot create a clone there.\n"); return; }
    int cnt = 0;
    for (int i = 0; i < nusers; i++) {
        struct user *c = users[i];
        if (c->type != T_CLONE || c->owner != u) continue;
        if (c->room == rm) { snprintf(text, sizeof(text), "You already have a clone in the %s.\n", rooms[rm].name); write_user(u, text); return; }
        if (++cnt == max_clones) { write_user(u, "You already have the maximum number of clones allowed.\n"); return; }
    }
    struct user *c = create_user();
    if (!c) return;
    c->type = T_CLONE; c->sock = u->sock; c->room = rm; c->owner = u;
    strcpy(c->name, u->name); strcpy(c->desc, "~BR(CLONE)");
    if (rm == u->room) write_user(u, "~FB~OLYou whisper a haunting spell and a clone is created here.\n");
    else { snprintf(text, sizeof(text), "~FB~OLYou whisper a haunting spell and a clone is created in the %s.\n", rooms[rm].name); write_user(u, text); }
    snprintf(text, sizeof(text), "~FB~OL%s whispers a haunting spell...\n", u->vis ? u->name : invisname);
    write_room_except(u->room, text, u);
    snprintf(text, sizeof(text), "~FB~OLA clone of %s appears in a swirling magical mist!\n", u->name);
    write_room_except(rm, text, u);
}

/* nuts333.c:7166-7210 */
static void destroy_clone(struct user *u)
{
    int rm = u->room;
    struct user *whose = u;
    if (word_count >= 2 && (rm = get_room(word[1])) < 0) { write_user(u, "There is no such room.\n"); return; }
    if (word_count > 2) {
        if (!(whose = get_user(word[2]))) { write_user(u, "There is no one of that name logged on.\n"); return; }
        if (whose->level >= u->level) { write_user(u, "You cannot destroy the clone of a user of an equal or higher level.\n"); return; }
    }
    struct user *c = find_clone(whose, rm);
    if (!c) {
        if (whose == u) snprintf(text, sizeof(text), "You do not have a clone in the %s.\n", rooms[rm].name);
        else snprintf(text, sizeof(text), "%s does not have a clone the %s.\n", whose->name, rooms[rm].name);
        write_user(u, text); return;
    }
    destruct_user(c);
    reset_access(rm);
    write_user(u, "~FM~OLYou whisper a sharp spell and the clone is destroyed.\n");
    snprintf(text, sizeof(text), "~FM~OL%s whispers a sharp spell...\n", u->vis ? u->name : invisname);
    write_room_except(u->room, text, u);
    snprintf(text, sizeof(text), "~FM~OLThe clone of %s shimmers and vanishes.\n", whose->name);
    write_room(rm, text);
    if (whose != u) { snprintf(text, sizeof(text), "~OLSYSTEM: ~FR%s has destroyed your clone in the %s.\n", u->name, rooms[rm].name); write_user(whose, text); }
    destructed = 0;
}

/* nuts333.c:7214-7234 */
static void myclones(struct user *u)
{
    int cnt = 0;
    for (int i = 0; i < nusers; i++) {
        struct user *c = users[i];
        if (c->type != T_CLONE || c->owner != u) continue;
        if (++cnt == 1) write_user(u, "\n~BB*** Rooms you have clones in ***\n\n");
        snprintf(text, sizeof(text), "  %s\n", rooms[c->room].name); write_user(u, text);
    }
    if (!cnt) { write_user(u, "You have no clones.\n"); return; }
    snprintf(text, sizeof(text), "\nTotal of %d clones.\n\n", cnt); write_user(u, text);
}

/* nuts333.c:7262-7291 */
static void clone_switch(struct user *u)
{
    if (word_count < 2) { write_user(u, "Usage: switch <room clone is in>\n"); return; }
    int rm = get_room(word[1]);
    if (rm < 0) { write_user(u, "There is no such room.\n"); return; }
    struct user *c = find_clone(u, rm);
    if (!c) { write_user(u, "You do not have a clone in that room.\n"); return; }
    write_user(u, "\n~FB~OLYou experience a strange sensation...\n");
    c->room = u->room; u->room = rm;
    snprintf(text, sizeof(text), "The clone of %s comes alive!\n", c->name); write_room_except(u->room, text, u);
    snprintf(text, sizeof(text), "%s turns into a clone!\n", c->name); write_room_except(c->room, text, c);
    look(u);
}

static void say(struct user *u, const char *inpstr);
/* nuts333.c:7295-7320 */
static void clone_say(struct user *u, const char *inpstr)
{
    if (u->muzzled) { write_user(u, "You are muzzled, your clone cannot speak.\n"); return; }
    if (word_count < 3) { write_user(u, "Usage: csay <room clone is in> <message>\n"); return; }
    int rm = get_room(word[1]);
    if (rm < 0) { write_user(u, "There is no such room.\n"); return; }
    struct user *c = find_clone(u, rm);
    if (!c) { write_user(u, "You do not have a clone in that room.\n"); return; }
    say(c, np_remove_first(inpstr));
}

/* nuts333.c:7325-7357 */
static void clone_hear(struct user *u)
{
    if (word_count < 3 || (strcmp(word[2], "all") && strcmp(word[2], "swears") && strcmp(word[2], "nothing"))) {
        write_user(u, "Usage: chear <room clone is in> all/swears/nothing\n"); return;
    }
    int rm = get_room(word[1]);
    if (rm < 0) { write_user(u, "There is no such room.\n"); return; }
    struct user *c = find_clone(u, rm);
    if (!c) { write_user(u, "You do not have a clone in that room.\n"); return; }
    if (!strcmp(word[2], "all")) { c->clone_hear = 2; write_user(u, "Clone will now hear everything.\n"); }
    else if (!strcmp(word[2], "swears")) { c->clone_hear = 1; write_user(u, "Clone will now only hear swearing.\n"); }
    else { c->clone_hear = 0; write_user(u, "Clone will now hear nothing.\n"); }
}

/* nuts333.c:2636-2642 */
static void cls(struct user *u) { for (int i = 0; i < 5; i++) write_user(u, "\n\n\n\n\n\n\n\n\n\n"); }

/* nuts333.c:6434-6456 */
static void visibility(struct user *u, int vis)
{
    if (vis) {
        if (u->vis) { write_user(u, "You are already visible.\n"); return; }
        write_user(u, "~FB~OLYou recite a melodic incantation and reappear.\n");
        snprintf(text, sizeof(text), "~FB~OLYou hear a melodic incantation chanted and %s materialises!\n", u->name);
        write_room_except(u->room, text, u);
        u->vis = 1; return;
    }
    if (!u->vis) { write_user(u, "You are already invisible.\n"); return; }
    write_user(u, "~FB~OLYou recite a melodic incantation and fade out.\n");
    snprintf(text, sizeof(text), "~FB~OL%s recites a melodic incantation and disappears!\n", u->name);
    write_room_except(u->room, text, u);
    u->vis = 0;
}

/* ------------------------------------------------------------------ dispatcher (L3) */
/* nuts333.c:3753-3937 */
static void exec_com(struct user *u, char *inpstr)
{
    com_num = -1;
    char *comword = word[0][0] == '.' ? word[0] + 1 : word[0];
    if (!comword[0]) { write_user(u, "Unknown command.\n"); return; }
    if (!strcmp(word[0], ">")) strcpy(word[0], "tell");
    if (!strcmp(word[0], "<")) strcpy(word[0], "pemote");
    if (!strcmp(word[0], "-")) strcpy(word[0], "echo");
    if (!strcmp(word[0], "!")) strcpy(word[0], "shout");
    if (inpstr[0] == ';') strcpy(word[0], "emote");
    else if (inpstr[0] == '#') strcpy(word[0], "semote");
    else inpstr = (char *)np_remove_first(inpstr);

    com_num = np_command_lookup(comword);
    if (u->room >= 0 && (com_num == -1 || np_command_level(com_num) > u->level)) { write_user(u, "Unknown command.\n"); return; }

    if (u->room < 0) {
        /* away over a netlink: a few commands run at home, the rest are relayed (c:3787-3806) */
        switch (com_num) {
        case NP_HOME: case NP_QUIT: case NP_MODE: case NP_PROMPT: case NP_COLOUR: case NP_REBOOT:
        case NP_SUICIDE: case NP_SHUTDOWN: case NP_CHARECHO:
            write_user(u, "~FY~OL*** Home execution ***\n"); break;
        default:
            snprintf(text, sizeof(text), "ACT %s %s %s\n", u->name, word[0], inpstr);
            write_sock(u->netlink->sock, text);
            no_prompt = 1;
            return;
        }
    }
    if (u->type == T_REMOTE) {
        switch (com_num) {
        case NP_PASSWD: case NP_ENTPRO: case NP_ACCREQ: case NP_CONN: case NP_DISCONN:
            write_user(u, "Sorry, remote users cannot use that command.\n"); return;
        default: break;
        }
    }

    switch (com_num) {
    case NP_QUIT: disconnect_user(u); break;
    case NP_LOOK: look(u); break;
    case NP_MODE:
        write_user(u, u->command_mode ? "Now in SPEECH mode.\n" : "Now in COMMAND mode.\n");      /* c:4009-4018 */
        u->command_mode = !u->command_mode; break;
    case NP_SAY:
        if (word_count < 2) { write_user(u, "Say what?\n"); return; }
        say(u, inpstr); break;
    case NP_SHOUT: shout(u, inpstr); break;
    case NP_TELL: tell(u, inpstr); break;
    case NP_EMOTE: emote(u, inpstr); break;
    case NP_SEMOTE: semote(u, inpstr); break;
    case NP_PEMOTE: pemote(u, inpstr); break;
    case NP_ECHO: echo(u, inpstr); break;
    case NP_GO: go(u); break;
    case NP_IGNALL:                                                                                   /* c:4463-4477 */
        if (!u->ignall) {
            write_user(u, "You are now ignoring everyone.\n");
            snprintf(text, sizeof(text), "%s is now ignoring everyone.\n", u->name);
        } else {
            write_user(u, "You will now hear everyone again.\n");
            snprintf(text, sizeof(text), "%s is listening again.\n", u->name);
        }
        write_room_except(u->room, text, u);
        u->ignall = !u->ignall; break;
    case NP_PROMPT:                                                                                   /* c:4481-4490 */
        write_user(u, u->prompt ? "Prompt ~FROFF.\n" : "Prompt ~FGON.\n");
        u->prompt = !u->prompt; break;
    case NP_REVIEW: review(u); break;
    case NP_VER: snprintf(text, sizeof(text), "NUTS version %s\n", VERSION); write_user(u, text); break;
    case NP_VIS: visibility(u, 1); break;
    case NP_INVIS: visibility(u, 0); break;
    case NP_COLOUR:                                                                                   /* c:7458-7481 */
        if (u->colour) { write_user(u, "Colour ~FROFF.\n"); u->colour = 0; }
        else { u->colour = 1; write_user(u, "Colour ~FGON.\n"); }
        if (u->room < 0) prompt(u);
        break;
    case NP_CHARECHO:                                                                                 /* c:6881-6893 */
        write_user(u, u->charmode_echo ? "Echoing for character mode clients ~FROFF.\n" : "Echoing for character mode clients ~FGON.\n");
        u->charmode_echo = !u->charmode_echo;
        if (u->room < 0) prompt(u);
        break;
    case NP_IGNSHOUT:                                                                                 /* c:7484-7494 */
        write_user(u, u->ignshout ? "You are no longer ignoring shouts and shout emotes.\n"
                                  : "You are now ignoring shouts and shout emotes.\n");
        u->ignshout = !u->ignshout; break;
    case NP_IGNTELL:                                                                                  /* c:7497-7507 */
        write_user(u, u->igntell ? "You are no longer ignoring tells and private emotes.\n"
                                 : "You are now ignoring tells and private emotes.\n");
        u->igntell = !u->igntell; break;
    case NP_REVTELL: revtell(u); break;
    case NP_BCAST: bcast(u, inpstr); break;
    case NP_WIZSHOUT: wizshout(u, inpstr); break;
    case NP_AFK: afk(u, inpstr); break;
    case NP_CLS: cls(u); break;
    case NP_CREATE: create_clone(u); break;
    case NP_DESTROY: destroy_clone(u); break;
    case NP_MYCLONES: myclones(u); break;
    case NP_SWITCH: clone_switch(u); break;
    case NP_CSAY: clone_say(u, inpstr); break;
    case NP_CHEAR: clone_hear(u); break;
    default:
        snprintf(text, sizeof(text), "[talker_port] '%s' is outside the restated path.\n", np_command_name(com_num));
        write_user(u, text);
    }
}

/* ------------------------------------------------------------------ input (L3/L4) */
/* nuts333.c:369-399 */
static int get_charclient_line(struct user *u, char *inpstr, int len)
{
    for (int l = 0; l < len; l++) {
        if (inpstr[l] == 8 || inpstr[l] == 127) {
            if (u->buffpos) { u->buffpos--; if (u->charmode_echo) write_user(u, "\b \b"); }
            continue;
        }
        u->buff[u->buffpos] = inpstr[l];
        if ((signed char)inpstr[l] < 32 || u->buffpos + 2 == NP_ARR_SIZE) {
            np_terminate(u->buff);
            strcpy(inpstr, u->buff);
            if (u->charmode_echo) write_user(u, "\n");
            return 1;
        }
        u->buffpos++;
    }
    if (u->charmode_echo && ((u->login != 2 && u->login != 1) || password_echo))
        if (write(u->sock, inpstr, (size_t)len) < 0) {}
    return 0;
}

/* one input chunk from one user: nuts333.c:136-235 */
static void user_input(struct user *u)
{
    char inpstr[NP_ARR_SIZE + 1];
    inpstr[0] = 0;
    int len = (int)read(u->sock, inpstr, NP_ARR_SIZE);
    if (len <= 0) { disconnect_user(u); return; }
    if ((unsigned char)inpstr[0] == 255) return;                      /* telnet IAC replies */
    if ((signed char)inpstr[len - 1] >= 32 || u->buffpos) {
        if (!get_charclient_line(u, inpstr, len)) return;
    } else {
        inpstr[len] = 0;
        np_terminate(inpstr);
    }
    no_prompt = 0; com_num = -1; force_listen = 0; destructed = 0;
    u->buff[0] = 0; u->buffpos = 0; u->last_input = time(NULL);
    if (u->login) { login(u, inpstr); return; }

    if (!strcmp(inpstr, ".") && u->inpstr_old[0]) {
        strcpy(inpstr, u->inpstr_old);
        snprintf(text, sizeof(text), "%s\n", inpstr);
        write_user(u, text);
    } else if (inpstr[0]) {
        strncpy(u->inpstr_old, inpstr, NP_REVIEW_LEN);
        u->inpstr_old[NP_REVIEW_LEN] = 0;
    }

    for (int w = 0; w < NP_MAX_WORDS; w++) word[w][0] = 0;
    word_count = np_wordfind(inpstr, word);
    if (u->afk) {
        if (u->afk == 2) {
            if (!word_count) { if (u->command_mode) prompt(u); return; }
            const char *h = crypt(word[0], "NU");
            if (!h || strcmp(h, u->pass)) { write_user(u, "Incorrect password.\n"); prompt(u); return; }
            cls(u);
            write_user(u, "Session unlocked, you are no longer AFK.\n");
        } else write_user(u, "You are no longer AFK.\n");
        u->afk_mesg[0] = 0;
        if (u->vis) { snprintf(text, sizeof(text), "%s comes back from being AFK.\n", u->name); write_room_except(u->room, text, u); }
        if (u->afk == 2) { u->afk = 0; prompt(u); return; }
        u->afk = 0;
    }
    if (!word_count) {
        if (u->room < 0) { snprintf(text, sizeof(text), "ACT %s NL\n", u->name); write_sock(u->netlink->sock, text); }
        if (u->command_mode) prompt(u);
        return;
    }
    com_num = -1;
    if (u->command_mode || strchr(".;!<>-#", inpstr[0])) exec_com(u, inpstr);
    else say(u, inpstr);
    if (destructed) return;
    if (u->room >= 0) prompt(u);
    else switch (com_num) {
        case -1: case NP_HOME: case NP_QUIT: case NP_MODE: case NP_PROMPT: case NP_SUICIDE: case NP_REBOOT: case NP_SHUTDOWN:
            prompt(u); break;
        default: break;
    }
}

/* nuts333.c:263-326 */
static void accept_connection(int lsock, int num);
static void accept_server_connection(int sock, struct sockaddr_in addr);
static void exec_netcom(struct netlink *nl, char *inpstr);
static void shutdown_netlink(struct netlink *nl);

static const char *site_of(struct sockaddr_in addr)
{
    static char site[81];
    snprintf(site, sizeof(site), "%s", inet_ntoa(addr.sin_addr));
    struct hostent *h = gethostbyaddr(&addr.sin_addr, 4, AF_INET);
    if (h) snprintf(site, sizeof(site), "%s", h->h_name);
    strtolower(site);
    return site;
}

static void accept_connection(int lsock, int num)
{
    struct sockaddr_in addr; socklen_t sz = sizeof(addr);
    int s = accept(lsock, (struct sockaddr *)&addr, &sz);
    if (s < 0) return;
    if (s >= FD_SETSIZE) {
        /* Deliberate divergence: the reference has no such guard (nuts333.c:94,251-258,274) and, once
           RLIMIT_NOFILE lets accept() hand out descriptor 1024, FD_SET() runs off the end of the mask: the
           fortified -O2 build aborts ("buffer overflow detected"), the -O0 build corrupts its stack and
           stops serving.  A test oracle must outlive its test, so we turn the connection away. */
        write_sock(s, "\n\rSorry, the talker is full at the moment.\n\n\r"); close(s); return;
    }
    if (num == 2) { accept_server_connection(s, addr); return; }
    char site[81]; snprintf(site, sizeof(site), "%s", site_of(addr));
    if (listed_in("siteban", site, 1)) { write_sock(s, "\n\rLogins from your site/domain are banned.\n\n\r"); close(s); return; }
    more(NULL, s, "motd1");
    if (num_of_users + num_of_logins >= max_users && !num) { write_sock(s, "\n\rSorry, the talker is full at the moment.\n\n\r"); close(s); return; }
    struct user *u = create_user();
    if (!u) { close(s); return; }
    u->sock = s; u->login = 3; u->last_input = time(NULL);
    u->port = port[num ? 1 : 0];
    if (num) write_user(u, "** Wizport login **\n\n");
    strcpy(u->site, site);
    u->site_port = ntohs(addr.sin_port);
    echo_on(u);
    write_user(u, "Give me a name: ");
    num_of_logins++;
}

/* ------------------------------------------------------------------ netlink (config #5) */
#include "talker_port_netlink.inc"

/* ------------------------------------------------------------------ boot */
static int yes_no(const char *w) { return !strcmp(w, "YES") ? 1 : !strcmp(w, "NO") ? 0 : -1; }
static int onoff(const char *w) { return !strcmp(w, "ON") ? 1 : !strcmp(w, "OFF") ? 0 : -1; }

/* nuts333.c:446-1008; options DOCS/about_config:11-174 */
static void load_config(void)
{
    char fn[128], line[82], w[8][81];
    snprintf(fn, sizeof(fn), "datafiles/%s", confile);
    FILE *fp = fopen(fn, "r");
    if (!fp) boot_exit("can't open config file %s", fn);
    int section = 0, lineno = 0;
    while (fgets(line, 81, fp)) {
        lineno++;
        for (int i = 0; i < 8; i++) w[i][0] = 0;
        sscanf(line, "%80s %80s %80s %80s %80s %80s %80s %80s", w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]);
        if (w[0][0] == '#' || !w[0][0]) continue;
        if (w[0][strlen(w[0]) - 1] == ':') {
            if (!strcmp(w[0], "INIT:")) section = 1;
            else if (!strcmp(w[0], "ROOMS:")) section = 2;
            else if (!strcmp(w[0], "SITES:")) section = 3;
            else boot_exit("unknown section header on line %d", lineno);
            continue;
        }
        if (section == 1) {
            if (!w[1][0]) boot_exit("required parameter missing on line %d", lineno);
            int v = atoi(w[1]), r = 0;
            if (!strcmp(w[0], "mainport")) port[0] = v;
            else if (!strcmp(w[0], "wizport")) port[1] = v;
            else if (!strcmp(w[0], "linkport")) port[2] = v;
            else if (!strcmp(w[0], "verification")) snprintf(verification, sizeof(verification), "%s", w[1]);
            else if (!strcmp(w[0], "max_users")) max_users = v;
            else if (!strcmp(w[0], "max_clones")) max_clones = v;
            else if (!strcmp(w[0], "min_private")) min_private_users = v;
            else if (!strcmp(w[0], "system_logging")) r = system_logging = onoff(w[1]);
            else if (!strcmp(w[0], "colour_def")) r = colour_def = onoff(w[1]);
            else if (!strcmp(w[0], "prompt_def")) r = prompt_def = onoff(w[1]);
            else if (!strcmp(w[0], "charecho_def")) r = charecho_def = onoff(w[1]);
            else if (!strcmp(w[0], "ban_swearing")) r = ban_swearing = yes_no(w[1]);
            else if (!strcmp(w[0], "auto_connect")) r = auto_connect = yes_no(w[1]);
            else if (!strcmp(w[0], "password_echo")) r = password_echo = yes_no(w[1]);
            else if (!strcmp(w[0], "allow_caps_in_name")) r = allow_caps_in_name = yes_no(w[1]);
            else if (!strcmp(w[0], "gatecrash_level")) r = gatecrash_level = get_level(w[1]);
            else if (!strcmp(w[0], "wizport_level")) r = wizport_level = get_level(w[1]);
            else if (!strcmp(w[0], "rem_user_maxlevel")) r = rem_user_maxlevel = get_level(w[1]);
            else if (!strcmp(w[0], "rem_user_deflevel")) r = rem_user_deflevel = get_level(w[1]);
            else if (!strcmp(w[0], "minlogin_level")) { minlogin_level = get_level(w[1]); if (minlogin_level < 0 && strcmp(w[1], "NONE")) r = -1; }
            else {
                /* accepted and ignored: they steer subsystems outside the path */
                static const char *ignored[] = { "mesg_life", "ignore_mp_level", "mesg_check_time", "heartbeat", "login_idle_time",
                    "user_idle_time", "ignore_sigterm", "crash_action", "time_out_afks", "time_out_maxlevel", NULL };
                int ok = 0;
                for (int i = 0; ignored[i]; i++) ok |= !strcmp(ignored[i], w[0]);
                if (!ok) boot_exit("unknown INIT option on line %d", lineno);
            }
            if (r < 0) boot_exit("bad value on line %d", lineno);
        } else if (section == 2) {
            if (!w[2][0]) boot_exit("required parameter(s) missing on line %d", lineno);
            if (nrooms == MAX_ROOMS) boot_exit("too many rooms");
            struct room *r = &rooms[nrooms++];
            memset(r, 0, sizeof(*r));
            snprintf(r->label, sizeof(r->label), "%s", w[0]);
            snprintf(r->name, sizeof(r->name), "%s", w[1]);
            char *save = NULL;
            for (char *t = strtok_r(w[2], ",", &save); t && r->nlinks < MAX_LINKS; t = strtok_r(NULL, ",", &save))
                snprintf(r->link_label[r->nlinks++], 6, "%s", t);
            if (w[3][0] == '#' || !w[3][0] || !strcmp(w[3], "BOTH")) r->access = ACC_PUBLIC;
            else if (!strcmp(w[3], "PUB")) r->access = ACC_FIXED_PUBLIC;
            else if (!strcmp(w[3], "PRIV")) r->access = ACC_FIXED_PRIVATE;
            else boot_exit("unknown room access type on line %d", lineno);
            if (w[3][0] != '#' && w[4][0] && w[4][0] != '#') {
                if (!strcmp(w[4], "ACCEPT")) r->inlink = 1;
                else if (!strcmp(w[4], "CONNECT") && w[5][0]) snprintf(r->netlink_name, sizeof(r->netlink_name), "%s", w[5]);
                else boot_exit("unknown connection option on line %d", lineno);
            }
        } else if (section == 3) {
            if (!w[3][0]) boot_exit("required parameter(s) missing on line %d", lineno);
            if (nnetlinks == MAX_NETLINKS) boot_exit("too many sites");
            struct netlink *nl = &netlinks[nnetlinks++];
            memset(nl, 0, sizeof(*nl));
            nl->in_use = 1; nl->sock = -1; nl->connect_room = -1;
            snprintf(nl->service, sizeof(nl->service), "%s", w[0]);
            strtolower(w[1]); snprintf(nl->site, sizeof(nl->site), "%s", w[1]);
            nl->port = atoi(w[2]);
            snprintf(nl->verification, sizeof(nl->verification), "%s", w[3]);
            nl->allow = !strcmp(w[4], "IN") ? ALLOW_IN : !strcmp(w[4], "OUT") ? ALLOW_OUT : ALLOW_ALL;
        } else boot_exit("section header expected on line %d", lineno);
    }
    fclose(fp);
    if (!verification[0] || !port[0] || !port[1] || !port[2] || !nrooms) boot_exit("incomplete config");
    for (int i = 0; i < nrooms; i++) {
        struct room *r = &rooms[i];
        for (int l = 0; l < r->nlinks; l++) {
            r->link[l] = -1;
            for (int j = 0; j < nrooms; j++) if (j != i && !strcmp(r->link_label[l], rooms[j].label)) { r->link[l] = j; break; }
            if (r->link[l] < 0) boot_exit("room %s has undefined link label '%s'", r->name, r->link_label[l]);
        }
        if (r->netlink_name[0]) {
            for (int n = 0; n < nnetlinks; n++) if (!strcmp(netlinks[n].service, r->netlink_name)) { r->nl = &netlinks[n]; break; }
            if (!r->nl) boot_exit("service name %s not defined for room %s", r->netlink_name, r->name);
        }
        snprintf(fn, sizeof(fn), "datafiles/%s.R", r->name);
        FILE *d = fopen(fn, "r");
        if (d) { size_t n = fread(r->desc, 1, sizeof(r->desc) - 1, d); r->desc[n] = 0; fclose(d); }
    }
}

static void init_sockets(void)
{
    for (int i = 0; i < 3; i++) {
        struct sockaddr_in a; memset(&a, 0, sizeof(a));
        a.sin_family = AF_INET; a.sin_addr.s_addr = INADDR_ANY; a.sin_port = htons((unsigned short)port[i]);
        if ((listen_sock[i] = socket(AF_INET, SOCK_STREAM, 0)) < 0) boot_exit("socket: %s", strerror(errno));
        int on = 1; setsockopt(listen_sock[i], SOL_SOCKET, SO_REUSEADDR, &on, sizeof(on));
        if (bind(listen_sock[i], (struct sockaddr *)&a, sizeof(a)) < 0) boot_exit("bind port %d: %s", port[i], strerror(errno));
        if (listen(listen_sock[i], 10) < 0) boot_exit("listen: %s", strerror(errno));
        fcntl(listen_sock[i], F_SETFL, O_NDELAY);
    }
}

int main(int argc, char **argv)
{
    if (argc > 1) snprintf(confile, sizeof(confile), "%s", argv[1]);
    fast_mode = getenv("NUTS_PORT_FAST") && getenv("NUTS_PORT_FAST")[0] == '1';
    printf("\n*** NUTS %s path restatement (talker_port) booting ***\n\n", VERSION);
    write_syslog("\n*** SERVER BOOTING ***\n", 0);
    signal(SIGPIPE, SIG_IGN);
    time_t now = time(NULL); struct tm *tm = localtime(&now); thour = tm->tm_hour; tmin = tm->tm_min;
    load_config();
    init_sockets();
    if (auto_connect) init_connections();
    /* same launch protocol as the reference (nuts333.c:79-87): the parent exits, the child
       announces its PID in ./syslog */
    fflush(stdout);
    switch (fork()) {
    case -1: boot_exit("fork failed");
    case 0: break;
    default: _exit(0);
    }
    snprintf(text, sizeof(text), "*** Booted successfully with PID %d ***\n\n", (int)getpid());
    write_syslog(text, 0);

    for (;;) {
        fd_set mask; FD_ZERO(&mask);
        for (int i = 0; i < 3; i++) FD_SET(listen_sock[i], &mask);
        for (int i = 0; i < nusers; i++) if (users[i]->type == T_LOCAL) FD_SET(users[i]->sock, &mask);
        for (int i = 0; i < nnetlinks; i++) if (netlinks[i].type != NL_UNCONNECTED) FD_SET(netlinks[i].sock, &mask);
        if (select(FD_SETSIZE, &mask, NULL, NULL, NULL) == -1) continue;

        for (int i = 0; i < 3; i++) if (FD_ISSET(listen_sock[i], &mask)) accept_connection(listen_sock[i], i);

        for (int i = 0; i < nnetlinks; i++) {
            struct netlink *nl = &netlinks[i];
            no_prompt = 0;
            if (nl->type == NL_UNCONNECTED || !FD_ISSET(nl->sock, &mask)) continue;
            char in[NP_ARR_SIZE];
            int len = (int)read(nl->sock, in, sizeof(in) - 3);
            if (len <= 0) {
                snprintf(text, sizeof(text), "~OLSYSTEM:~RS Lost link to %s in the %s.\n", nl->service,
                         nl->connect_room >= 0 ? rooms[nl->connect_room].name : "?");
                write_room(-1, text);
                shutdown_netlink(nl);
                continue;
            }
            in[len] = 0;
            exec_netcom(nl, in);
        }

        /* users may be created or destroyed while we walk: re-find our place by pointer */
        for (int i = 0; i < nusers;) {
            struct user *u = users[i];
            if (u->type == T_LOCAL && u->sock >= 0 && u->sock < FD_SETSIZE && FD_ISSET(u->sock, &mask)) {
                FD_CLR(u->sock, &mask);
                user_input(u);
                int at = user_index(u);
                i = at >= 0 ? at + 1 : (i < nusers && users[i] != u ? i : i);
                if (at < 0) continue;      /* u is gone; users[i] is now its successor */
            } else i++;
        }
    }
}
