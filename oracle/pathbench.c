/*
 * pathbench.c -- how long does the USER-SPACE part of one delivered line take?
 *
 * TEST INFRASTRUCTURE (see nuts_path.h).  Times the restated hot-path functions in a tight
 * loop with the syscall replaced by a memcpy, so the number is the work a device could at
 * best take over: transducing one broadcast line for one recipient (nuts333.c:1315-1365),
 * the six-term fan-out predicate (nuts333.c:1410-1415), and formatting the line once
 * (nuts333.c:4094-4097).  /proc/<pid>/stat's 10 ms ticks are too coarse for this.
 *
 *   pathbench [iterations]      -> one JSON object
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "nuts_path.h"

static char sink[4096]; static size_t sink_len;
static void emit(void *ctx, const char *buf, size_t len) { (void)ctx; memcpy(sink, buf, len); sink_len += len; }

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec * 1e9 + (double)t.tv_nsec; }

int main(int argc, char **argv)
{
    long n = argc > 1 ? atol(argv[1]) : 5000000;
    char say[256], shout[256], text[2048];
    snprintf(say, sizeof(say), "Uaaa says: synthetic broadcast line %06d from the nuts333 bench\n", 123);
    snprintf(shout, sizeof(shout), "~OLUaaa shouts:~RS synthetic broadcast line %06d from the nuts333 bench\n", 123);
    struct { const char *name, *str; int colour; } cases[] = {
        { "say_colour_off", say, 0 }, { "say_colour_on", say, 1 }, { "shout_colour_off", shout, 0 }, { "shout_colour_on", shout, 1 } };
    printf("{\"iterations\":%ld", n);
    for (unsigned c = 0; c < 4; c++) {
        double t0 = now();
        for (long i = 0; i < n; i++) np_write_user_stream(cases[c].str, cases[c].colour, emit, NULL);
        printf(",\"transduce_%s_ns\":%.1f", cases[c].name, (now() - t0) / (double)n);
    }
    struct np_listener l = { 0, 1, 1, 0, 0, 0 }; volatile int admitted = 0;
    double t0 = now();
    for (long i = 0; i < n; i++) { l.ignshout = (int)(i & 1); admitted += np_fanout_admits(&l, 0, 0, NP_SAY); }
    printf(",\"fanout_predicate_ns\":%.2f", (now() - t0) / (double)n);
    t0 = now();
    for (long i = 0; i < n; i++) {
        const char *in = "synthetic broadcast line 000123 from the nuts333 bench";
        snprintf(text, sizeof(text), "%s %ss: %s\n", "Uaaa", np_say_verb(in), in);
    }
    printf(",\"format_line_once_ns\":%.1f", (now() - t0) / (double)n);
    printf(",\"sink\":%zu,\"admitted\":%d}\n", sink_len & 1, admitted & 1);
    return 0;
}
