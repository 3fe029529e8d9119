/*
 * nuts_path.h -- CPU restatement of the NUTS 3.3.3 input -> broadcast path.
 *
 * TEST INFRASTRUCTURE.  This is the oracle: a from-scratch restatement of the functions
 * SURVEY.md section 8(a) lists, used only by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg.  Nothing here is a product path (the project has none: BASELINE.json's
 * north_star flags the reference as not graft-eligible).
 *
 * Parity pin: byte-exact against tests/golden/ *.json, which are transcripts captured from
 * the unmodified reference build (oracle/_ref/nuts333, -O0 and -O2 agree) by
 * tests/golden/make_golden.py.
 *
 * Every function cites the reference lines it restates (paths relative to /root/reference).
 */
#ifndef NUTS_PATH_H
#define NUTS_PATH_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NP_ARR_SIZE 1000      /* nuts333.h:19  input line / read() size        */
#define NP_OUT_BUFF 1000      /* nuts333.h:16  write_user staging buffer       */
#define NP_MAX_WORDS 10       /* nuts333.h:17                                  */
#define NP_WORD_LEN 40        /* nuts333.h:18                                  */
#define NP_REVIEW_LEN 200     /* nuts333.h:39                                  */
#define NP_REVIEW_LINES 15    /* nuts333.h:37                                  */
#define NP_REVTELL_LINES 5    /* nuts333.h:38                                  */
#define NP_NUM_COLS 21        /* nuts333.h:21                                  */
#define NP_TEXT_SIZE (NP_ARR_SIZE * 2)   /* nuts333.h:280 global text[]         */

enum np_level { NP_NEW, NP_USER, NP_WIZ, NP_ARCH, NP_GOD };              /* nuts333.h:51-55 */

/* command numbers: index into the command table, nuts333.h:157-201 */
enum np_com {
    NP_QUIT, NP_LOOK, NP_MODE, NP_SAY, NP_SHOUT, NP_TELL, NP_EMOTE, NP_SEMOTE, NP_PEMOTE, NP_ECHO,
    NP_GO, NP_IGNALL, NP_PROMPT, NP_DESC, NP_INPHRASE, NP_OUTPHRASE, NP_PUBCOM, NP_PRIVCOM, NP_LETMEIN,
    NP_INVITE, NP_TOPIC, NP_MOVE, NP_BCAST, NP_WHO, NP_PEOPLE, NP_HELP, NP_SHUTDOWN, NP_NEWS, NP_READ,
    NP_WRITE, NP_WIPE, NP_SEARCH, NP_REVIEW, NP_HOME, NP_STATUS, NP_VER, NP_RMAIL, NP_SMAIL, NP_DMAIL,
    NP_FROM, NP_ENTPRO, NP_EXAMINE, NP_RMST, NP_RMSN, NP_NETSTAT, NP_NETDATA, NP_CONN, NP_DISCONN,
    NP_PASSWD, NP_KILL, NP_PROMOTE, NP_DEMOTE, NP_LISTBANS, NP_BAN, NP_UNBAN, NP_VIS, NP_INVIS, NP_SITE,
    NP_WAKE, NP_WIZSHOUT, NP_MUZZLE, NP_UNMUZZLE, NP_MAP, NP_LOGGING, NP_MINLOGIN, NP_SYSTEM, NP_CHARECHO,
    NP_CLEARLINE, NP_FIX, NP_UNFIX, NP_VIEWLOG, NP_ACCREQ, NP_REVCLR, NP_CREATE, NP_DESTROY, NP_MYCLONES,
    NP_ALLCLONES, NP_SWITCH, NP_CSAY, NP_CHEAR, NP_RSTAT, NP_SWBAN, NP_AFK, NP_CLS, NP_COLOUR, NP_IGNSHOUT,
    NP_IGNTELL, NP_SUICIDE, NP_DELETE, NP_REBOOT, NP_RECOUNT, NP_REVTELL, NP_NUM_COMMANDS
};

/* ---- input framing (nuts333.c:403-411, 417-432, 2350-2358) ---- */
int np_terminate(char *str);
int np_wordfind(const char *inpstr, char words[NP_MAX_WORDS][NP_WORD_LEN + 1]);
const char *np_remove_first(const char *inpstr);

/* ---- command table (nuts333.h:157-226, nuts333.c:3776-3781) ---- */
int np_command_count(void);
const char *np_command_name(int com);
int np_command_level(int com);
int np_command_lookup(const char *comword);

/* ---- colour-markup transducer (nuts333.c:1315-1365, 2562-2610) ---- */
typedef void (*np_emit_fn)(void *ctx, const char *buf, size_t len);
/* The 1000-byte staging buffer of write_user / more() (nuts333.c:1296, 2211), kept across strings by more(). */
struct np_stage { char buff[NP_OUT_BUFF + 8]; int pos; };
void np_stage_init(struct np_stage *st);
void np_stage_feed(struct np_stage *st, const char *str, int colour, np_emit_fn emit, void *ctx);
void np_stage_flush(struct np_stage *st, np_emit_fn emit, void *ctx);
/* Calls emit() once per write(2) the reference issues, with the same bytes. */
void np_write_user_stream(const char *str, int colour, np_emit_fn emit, void *ctx);
/* Concatenation of those chunks; returns the length needed (may exceed cap; no NUL). */
size_t np_transduce(const char *str, int colour, char *out, size_t cap);
/* Number of write(2) calls the reference makes for this string. */
int np_write_count(const char *str, int colour);
size_t np_colour_com_strip(const char *str, char *out, size_t cap);
const char *np_colour_code(int i);     /* ANSI sequence i, nuts333.h:237-246 */
const char *np_colour_com(int i);      /* two-letter command i, nuts333.h:249-255 */

/* ---- speech formatting (nuts333.c:4080-4097, 4119-4122, 4174-4179, 4202-4203, 4223-4224, 4276-4278) ---- */
const char *np_say_verb(const char *inpstr);
int np_contains_swearing(const char *str);                       /* nuts333.c:2540-2559 */

/* ---- fan-out predicate (nuts333.c:1410-1415) ---- */
struct np_listener {
    int login;              /* still in the login FSM                      */
    int has_room;           /* room != NULL (NULL == away over a netlink)  */
    int same_room;          /* room == rm                                  */
    int ignall, ignshout;
    int is_sender;
};
int np_fanout_admits(const struct np_listener *u, int rm_is_null, int force_listen, int com_num);

/* ---- review rings (nuts333.c:2062-2082) ---- */
void np_record(char *ring, int nlines, int *revline, const char *str);   /* ring: nlines x (NP_REVIEW_LEN+2) */

#ifdef __cplusplus
}
#endif
#endif
