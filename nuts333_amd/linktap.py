"""A counting relay for one talker<->talker netlink connection.

BASELINE configuration #5 asks for the ``MSG ... EMSG`` frames that cross the link to be *counted*
(SURVEY.md section 8d.5).  The dialling talker is pointed at this relay instead of its peer's link
port; the relay connects onward and copies bytes both ways unchanged, splitting each direction's
stream into newline-terminated lines and counting them by first word -- the 21 netlink verbs
(``nuts333.c:2956-2962``) plus the free-text body lines that travel between ``MSG`` and ``EMSG``
(``nuts333.c:1302-1305``).

It is a measuring instrument for tests and for the frame-count column of the baseline; the timed
runs leave it out (``workloads.config5(tap=False)``) because one more hop changes the Nagle /
delayed-ACK rhythm that bounds that configuration, and use the talkers' own ``write(2)`` counts
(``/proc/<pid>/io``) instead.  Both methods are asserted equal in ``tests/test_harness.py``.
"""
from __future__ import annotations

import socket
import threading
from collections import Counter

VERBS = ("DISCONNECT", "TRANS", "REL", "ACT", "GRANTED", "DENIED", "MSG", "EMSG", "PRM", "VERIFICATION",
         "VERIFY", "REMVD", "ERROR", "EXISTS?", "EXISTS_NO", "EXISTS_YES", "MAIL", "ENDMAIL", "MAILERROR",
         "KA", "RSTAT", "NUTS")


class LinkTap:
    """Listen on a free loopback port; relay the single connection accepted there to ``target_port``."""

    def __init__(self, target_port: int, host: str = "127.0.0.1", marker: bytes = b""):
        self.target = (host, target_port)
        #: frames whose text contains ``marker`` are also counted separately (``marked``): the workload's payload
        #: carries a fixed phrase, which tells the timed lines from the hand-shake and placement traffic
        self.marker = marker
        self.marked = [Counter(), Counter()]
        self.lsock = socket.socket()
        self.lsock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        self.lsock.bind((host, 0))
        self.lsock.listen(1)
        self.port = self.lsock.getsockname()[1]
        #: lines seen from the dialler towards the acceptor / from the acceptor back, keyed by verb
        #: ("(body)" for the text lines inside a MSG..EMSG or MAIL..ENDMAIL frame)
        self.dial_to_accept: Counter[str] = Counter()
        self.accept_to_dial: Counter[str] = Counter()
        self.bytes = [0, 0]
        self._threads: list[threading.Thread] = []
        self._socks: list[socket.socket] = []
        self._lock = threading.Lock()
        t = threading.Thread(target=self._serve, daemon=True)
        t.start()
        self._threads.append(t)

    def _serve(self) -> None:
        try:
            a, _ = self.lsock.accept()
        except OSError:
            return
        b = socket.create_connection(self.target)
        for s in (a, b):
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        self._socks += [a, b]
        for src, dst, ctr, k in ((a, b, self.dial_to_accept, 0), (b, a, self.accept_to_dial, 1)):
            t = threading.Thread(target=self._pump, args=(src, dst, ctr, k), daemon=True)
            t.start()
            self._threads.append(t)

    def _pump(self, src: socket.socket, dst: socket.socket, ctr: Counter, k: int) -> None:
        pending = b""
        in_body = False
        frame_marked = False
        while True:
            try:
                data = src.recv(65536)
            except OSError:
                break
            if not data:
                break
            try:
                dst.sendall(data)
            except OSError:
                break
            pending += data
            *lines, pending = pending.split(b"\n")
            with self._lock:
                self.bytes[k] += len(data)
                for ln in lines:
                    word = ln.split(b" ", 1)[0].decode("latin-1")
                    hit = bool(self.marker) and self.marker in ln
                    if in_body and word not in ("EMSG", "ENDMAIL"):
                        ctr["(body)"] += 1
                        frame_marked = frame_marked or hit
                        continue
                    ctr[word if word in VERBS else "(other)"] += 1
                    if word in ("MSG", "MAIL"):
                        in_body, frame_marked = True, False
                    elif word in ("EMSG", "ENDMAIL"):
                        in_body = False
                        if frame_marked:
                            self.marked[k]["MSG" if word == "EMSG" else "MAIL"] += 1
                    elif hit:
                        self.marked[k][word] += 1
        for s in (src, dst):
            try:
                s.shutdown(socket.SHUT_RDWR)
            except OSError:
                pass

    def snapshot(self) -> dict:
        with self._lock:
            return {"dial_to_accept": dict(self.dial_to_accept), "accept_to_dial": dict(self.accept_to_dial),
                    "marked_dial_to_accept": dict(self.marked[0]), "marked_accept_to_dial": dict(self.marked[1]),
                    "bytes_dial_to_accept": self.bytes[0], "bytes_accept_to_dial": self.bytes[1]}

    def close(self) -> None:
        for s in [self.lsock] + self._socks:
            try:
                s.close()
            except OSError:
                pass
