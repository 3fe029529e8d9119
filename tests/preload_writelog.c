/* TEST INFRASTRUCTURE: an LD_PRELOAD shim that logs the size of every write(2) a talker issues on a socket.
 *
 *   gcc -O2 -fPIC -shared tests/preload_writelog.c -o <tmp>/writelog.so -ldl
 *   LD_PRELOAD=<tmp>/writelog.so WRITELOG=<file> ./talker config
 *
 * strace is not available in this image and TCP coalesces segments, so the byte stream a client receives cannot
 * show where one write(2) ended and the next began; this can.  Used by tests/test_harness.py to check that the
 * restatement's more() / write_user() flush their 1000-byte staging buffer exactly where the reference does
 * (nuts333.c:1315-1365, 2250-2296).  One line per call: "<pid> <fd> <length>".
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/stat.h>
#include <unistd.h>

ssize_t write(int fd, const void *buf, size_t n)
{
    static ssize_t (*real)(int, const void *, size_t);
    if (!real) real = (ssize_t (*)(int, const void *, size_t))dlsym(RTLD_NEXT, "write");
    const char *path = getenv("WRITELOG");
    struct stat st;
    if (path && fstat(fd, &st) == 0 && S_ISSOCK(st.st_mode)) {
        int lf = open(path, O_WRONLY | O_APPEND | O_CREAT, 0644);
        if (lf >= 0) {
            char line[64];
            int l = snprintf(line, sizeof(line), "%d %d %zu\n", (int)getpid(), fd, n);
            if (real(lf, line, (size_t)l) < 0) {}
            close(lf);
        }
    }
    return real(fd, buf, n);
}
