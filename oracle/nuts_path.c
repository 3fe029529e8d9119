/*
 * nuts_path.c -- CPU restatement of the NUTS 3.3.3 input -> broadcast path (pure functions).
 *
 * TEST INFRASTRUCTURE / ORACLE -- see nuts_path.h.  Restated function by function from the
 * lines of the reference each one cites (file:line in the comment above it), with our own
 * data layout, re-entrant signatures and bounded buffers; the control flow of the byte
 * transducer (np_stage_feed) is our own arrangement of the reference's rules.  Where the
 * reference has an observable quirk the quirk is kept and called out, because the golden
 * transcripts (captured from the real thing) contain it.
 */
#include "nuts_path.h"

#include <ctype.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ input framing */

/* nuts333.c:403-411.  Cut the line at the first byte whose value, read as a SIGNED char,
 * is below 32 -- so every control character and every byte >= 0x80 ends the line.  A line
 * with no such byte in its first NP_ARR_SIZE characters is cut at NP_ARR_SIZE-1.
 * Returns the resulting length. */
int np_terminate(char *str)
{
    for (int i = 0; i < NP_ARR_SIZE; i++) {
        if ((signed char)str[i] < 32) { str[i] = 0; return i; }
    }
    str[NP_ARR_SIZE - 1] = 0;
    return NP_ARR_SIZE - 1;
}

/* nuts333.c:417-432.  Split into at most NP_MAX_WORDS words of at most NP_WORD_LEN-1
 * characters.  Quirks kept: a longer word spills into the next slot; when all ten slots
 * fill up the function reports nine. */
int np_wordfind(const char *in, char words[NP_MAX_WORDS][NP_WORD_LEN + 1])
{
    int wn = 0;
    while (wn < NP_MAX_WORDS) {
        while ((signed char)*in < 33) {
            if (!*in) return wn;
            in++;
        }
        int wpos = 0;
        while ((signed char)*in > 32 && wpos < NP_WORD_LEN - 1) words[wn][wpos++] = *in++;
        words[wn][wpos] = 0;
        wn++;
    }
    return wn - 1;
}

/* nuts333.c:2350-2358.  Position of the second word. */
const char *np_remove_first(const char *s)
{
    while ((signed char)*s < 33 && *s) s++;
    while ((signed char)*s > 32) s++;
    while ((signed char)*s < 33 && *s) s++;
    return s;
}

/* ------------------------------------------------------------------ command table */

/* names nuts333.h:157-177, minimum levels nuts333.h:206-226 (same order) */
static const struct { const char *name; unsigned char level; } COMMANDS[NP_NUM_COMMANDS] = {
    {"quit", NP_NEW}, {"look", NP_NEW}, {"mode", NP_NEW}, {"say", NP_NEW}, {"shout", NP_USER},
    {"tell", NP_USER}, {"emote", NP_USER}, {"semote", NP_USER}, {"pemote", NP_USER}, {"echo", NP_USER},
    {"go", NP_USER}, {"ignall", NP_USER}, {"prompt", NP_NEW}, {"desc", NP_USER}, {"inphr", NP_USER},
    {"outphr", NP_USER}, {"public", NP_USER}, {"private", NP_USER}, {"letmein", NP_USER}, {"invite", NP_USER},
    {"topic", NP_USER}, {"move", NP_WIZ}, {"bcast", NP_WIZ}, {"who", NP_NEW}, {"people", NP_WIZ},
    {"help", NP_NEW}, {"shutdown", NP_GOD}, {"news", NP_USER}, {"read", NP_NEW}, {"write", NP_USER},
    {"wipe", NP_WIZ}, {"search", NP_USER}, {"review", NP_USER}, {"home", NP_USER}, {"status", NP_NEW},
    {"version", NP_NEW}, {"rmail", NP_NEW}, {"smail", NP_USER}, {"dmail", NP_USER}, {"from", NP_USER},
    {"entpro", NP_USER}, {"examine", NP_USER}, {"rmst", NP_NEW}, {"rmsn", NP_NEW}, {"netstat", NP_WIZ},
    {"netdata", NP_ARCH}, {"connect", NP_GOD}, {"disconnect", NP_GOD}, {"passwd", NP_USER}, {"kill", NP_ARCH},
    {"promote", NP_WIZ}, {"demote", NP_WIZ}, {"listbans", NP_WIZ}, {"ban", NP_ARCH}, {"unban", NP_ARCH},
    {"vis", NP_ARCH}, {"invis", NP_ARCH}, {"site", NP_WIZ}, {"wake", NP_USER}, {"wizshout", NP_WIZ},
    {"muzzle", NP_WIZ}, {"unmuzzle", NP_WIZ}, {"map", NP_USER}, {"logging", NP_GOD}, {"minlogin", NP_GOD},
    {"system", NP_WIZ}, {"charecho", NP_NEW}, {"clearline", NP_ARCH}, {"fix", NP_GOD}, {"unfix", NP_GOD},
    {"viewlog", NP_WIZ}, {"accreq", NP_NEW}, {"revclr", NP_USER}, {"clone", NP_ARCH}, {"destroy", NP_ARCH},
    {"myclones", NP_ARCH}, {"allclones", NP_USER}, {"switch", NP_ARCH}, {"csay", NP_ARCH}, {"chear", NP_ARCH},
    {"rstat", NP_WIZ}, {"swban", NP_ARCH}, {"afk", NP_USER}, {"cls", NP_NEW}, {"colour", NP_NEW},
    {"ignshout", NP_USER}, {"igntell", NP_USER}, {"suicide", NP_NEW}, {"delete", NP_GOD}, {"reboot", NP_GOD},
    {"recount", NP_GOD}, {"revtell", NP_USER},
};

int np_command_count(void) { return NP_NUM_COMMANDS; }
const char *np_command_name(int com) { return (com >= 0 && com < NP_NUM_COMMANDS) ? COMMANDS[com].name : NULL; }
int np_command_level(int com) { return (com >= 0 && com < NP_NUM_COMMANDS) ? COMMANDS[com].level : -1; }

/* nuts333.c:3776-3781.  First table entry that begins with comword wins, so "s" is say,
 * "sh" is shout, "se" is semote.  An empty comword is handled by the caller (c:3763). */
int np_command_lookup(const char *comword)
{
    size_t len = strlen(comword);
    for (int i = 0; i < NP_NUM_COMMANDS; i++)
        if (!strncmp(COMMANDS[i].name, comword, len)) return i;
    return -1;
}

/* ------------------------------------------------------------------ colour markup */

static const char *const COLCODE[NP_NUM_COLS] = {            /* nuts333.h:237-246 */
    "\033[0m", "\033[1m", "\033[4m", "\033[5m", "\033[7m",
    "\033[30m", "\033[31m", "\033[32m", "\033[33m", "\033[34m", "\033[35m", "\033[36m", "\033[37m",
    "\033[40m", "\033[41m", "\033[42m", "\033[43m", "\033[44m", "\033[45m", "\033[46m", "\033[47m",
};
static const char COLCOM[NP_NUM_COLS][3] = {                  /* nuts333.h:249-255 */
    "RS", "OL", "UL", "LI", "RV", "FK", "FR", "FG", "FY", "FB", "FM", "FT", "FW",
    "BK", "BR", "BG", "BY", "BB", "BM", "BT", "BW",
};

const char *np_colour_code(int i) { return (i >= 0 && i < NP_NUM_COLS) ? COLCODE[i] : NULL; }
const char *np_colour_com(int i) { return (i >= 0 && i < NP_NUM_COLS) ? COLCOM[i] : NULL; }

/* index of the two-letter colour command at p, or -1 (never reads past a NUL) */
static int colcom_at(const char *p)
{
    if (!p[0] || !p[1]) return -1;
    for (int i = 0; i < NP_NUM_COLS; i++)
        if (p[0] == COLCOM[i][0] && p[1] == COLCOM[i][1]) return i;
    return -1;
}

/* nuts333.c:1315-1365.  The per-recipient byte transducer:
 *   '\n'        -> [ESC[0m if colour] "\n\r"
 *   "/~"        -> "~" (escape: the slash is dropped, the tilde is literal)
 *   "~XX"       -> ANSI code XX if colour, nothing otherwise (XX one of 21 commands)
 *   other '~'   -> "~"
 * staged through a 1000-byte buffer that is flushed (= one write(2)) when it is full, or
 * before a newline / tilde once fewer than 6 bytes remain; a trailing ESC[0m goes out as a
 * separate write when colour is on. */
void np_stage_init(struct np_stage *st) { st->pos = 0; }

/* One string through the staging buffer WITHOUT the final flush: what is left stays staged for the next
 * call.  write_user (below) stages one string and flushes; the file pager more() stages every line of a
 * file through the same buffer and flushes once at the end (nuts333.c:2250-2296), so its write(2)
 * boundaries fall where the running position says, not per line.  The "/~" look-behind is per string in
 * both (start of str / start of the line buffer, nuts333.c:1334, 2264). */
void np_stage_feed(struct np_stage *st, const char *str, int colour, np_emit_fn emit, void *ctx)
{
    char *buff = st->buff;
    int pos = st->pos;
    const char *const start = str;

    while (*str) {
        if (*str == '\n') {
            if (pos > NP_OUT_BUFF - 6) { emit(ctx, buff, (size_t)pos); pos = 0; }
            if (colour) { memcpy(buff + pos, "\033[0m", 4); pos += 4; }
            buff[pos++] = '\n';
            buff[pos++] = '\r';
            str++;
        } else if (str[0] == '/' && str[1] == '~') {
            str++;                      /* drop the slash; no fullness check on this path */
            continue;
        } else if (str != start && str[0] == '~' && str[-1] == '/') {
            buff[pos++] = '~';
            str++;
        } else if (*str == '~') {
            if (pos > NP_OUT_BUFF - 6) { emit(ctx, buff, (size_t)pos); pos = 0; }
            int c = colcom_at(str + 1);
            if (c >= 0) {
                if (colour) { size_t l = strlen(COLCODE[c]); memcpy(buff + pos, COLCODE[c], l); pos += (int)l; }
                str += 3;
            } else {
                buff[pos++] = '~';
                str++;
            }
        } else {
            buff[pos++] = *str++;
        }
        if (pos == NP_OUT_BUFF) { emit(ctx, buff, NP_OUT_BUFF); pos = 0; }
    }
    st->pos = pos;
}

void np_stage_flush(struct np_stage *st, np_emit_fn emit, void *ctx)
{
    if (st->pos) emit(ctx, st->buff, (size_t)st->pos);
    st->pos = 0;
}

void np_write_user_stream(const char *str, int colour, np_emit_fn emit, void *ctx)
{
    struct np_stage st;
    np_stage_init(&st);
    np_stage_feed(&st, str, colour, emit, ctx);
    np_stage_flush(&st, emit, ctx);
    if (colour) emit(ctx, "\033[0m", 4);
}

struct collect { char *out; size_t cap, len; int writes; };
static void collect_emit(void *ctx, const char *buf, size_t len)
{
    struct collect *c = ctx;
    if (c->out && c->len < c->cap) {
        size_t n = len < c->cap - c->len ? len : c->cap - c->len;
        memcpy(c->out + c->len, buf, n);
    }
    c->len += len;
    c->writes++;
}

size_t np_transduce(const char *str, int colour, char *out, size_t cap)
{
    struct collect c = { out, cap, 0, 0 };
    np_write_user_stream(str, colour, collect_emit, &c);
    return c.len;
}

int np_write_count(const char *str, int colour)
{
    struct collect c = { NULL, 0, 0, 0 };
    np_write_user_stream(str, colour, collect_emit, &c);
    return c.writes;
}

/* nuts333.c:2588-2610.  Remove every "~XX"; note that, unlike the transducer, this one
 * knows nothing about the "/~" escape. */
size_t np_colour_com_strip(const char *s, char *out, size_t cap)
{
    size_t n = 0;
    while (*s) {
        if (*s == '~' && colcom_at(s + 1) >= 0) { s += 3; continue; }
        if (n + 1 < cap) out[n] = *s;
        n++; s++;
    }
    if (cap) out[n < cap ? n : cap - 1] = 0;
    return n;
}

/* ------------------------------------------------------------------ speech helpers */

/* nuts333.c:4080-4084 */
const char *np_say_verb(const char *inpstr)
{
    size_t l = strlen(inpstr);
    char last = l ? inpstr[l - 1] : 0;
    return last == '?' ? "ask" : last == '!' ? "exclaim" : "say";
}

/* nuts333.c:2540-2559 with the word list of nuts333.h:275-277: case-insensitive substring. */
int np_contains_swearing(const char *str)
{
    static const char *const words[] = { "fuck", "shit", "cunt" };
    size_t l = strlen(str);
    char *low = malloc(l + 1);
    if (!low) return 0;
    for (size_t i = 0; i <= l; i++) low[i] = (char)tolower((unsigned char)str[i]);
    int hit = 0;
    for (size_t w = 0; w < sizeof(words) / sizeof(words[0]) && !hit; w++) hit = strstr(low, words[w]) != NULL;
    free(low);
    return hit;
}

/* ------------------------------------------------------------------ fan-out predicate */

/* nuts333.c:1410-1415.  A listener is skipped when it is still logging in, is away over a
 * netlink, is in another room (unless the message is for every room), ignores everything
 * (unless force_listen), ignores shouts and this is a shout/shout-emote, or is the sender. */
int np_fanout_admits(const struct np_listener *u, int rm_is_null, int force_listen, int com_num)
{
    if (u->login) return 0;
    if (!u->has_room) return 0;
    if (!u->same_room && !rm_is_null) return 0;
    if (u->ignall && !force_listen) return 0;
    if (u->ignshout && (com_num == NP_SHOUT || com_num == NP_SEMOTE)) return 0;
    if (u->is_sender) return 0;
    return 1;
}

/* ------------------------------------------------------------------ review rings */

/* nuts333.c:2062-2070 / 2074-2082.  Keep the first NP_REVIEW_LEN characters; a line that
 * long loses its own newline, so one is forced at position NP_REVIEW_LEN. */
void np_record(char *ring, int nlines, int *revline, const char *str)
{
    char *slot = ring + (size_t)*revline * (NP_REVIEW_LEN + 2);
    strncpy(slot, str, NP_REVIEW_LEN);
    slot[NP_REVIEW_LEN] = '\n';
    slot[NP_REVIEW_LEN + 1] = 0;
    *revline = (*revline + 1) % nlines;
}
