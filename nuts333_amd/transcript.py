"""Deterministic scripted telnet clients: capture byte-exact wire transcripts from a talker.

A *scenario* is a list of steps executed strictly one at a time (the talker drops
pipelined input, ``nuts333.c:136,149,403-411``).  After every step each connected,
logged-in client performs a ``.version`` round trip (``nuts333.c:3870-3872``): TCP keeps
per-socket order and the talker is single-threaded, so once a client has seen the version
reply it has also seen every byte the step caused to be written to it.  The reply itself
is cut off the capture.  No "quiet for N ms" heuristics -- with one exception: multi-segment
``raw`` steps pause between segments so the talker read()s each one on its own.

Only three things in a transcript are not a pure function of (accounts, script):
the peer ``(site:port)`` shown to WIZ+ on sign-on (``nuts333.c:1732``), the wall clock
in the prompt (``nuts333.c:2191-2195``) and the date stamp on board messages and mail
(``long_date(0)`` and the raw ``time_t`` beside it, ``nuts333.c:2614, 5025-5027, 2469, 2493-2496``); :func:`normalise` masks exactly those.
"""
from __future__ import annotations

import re
import select
import socket
import time
from dataclasses import dataclass, field
from typing import Iterable

VERSION_LINE = b"NUTS version 3.3.3"
RESET = b"\x1b[0m"

_SITE_PORT = re.compile(rb"\([A-Za-z0-9_.\-]+:\d{1,5}\)")
_PROMPT = re.compile(rb"<\d\d:\d\d, \d\d:\d\d, ")
# long_date(0), nuts333.c:2614-2623: the stamp on board messages and mail ("[ Sunday 4th October 2026 at 13:05 ]"-like)
# a time_t followed by a bare CR: the machine-readable stamp in front of a board header ("PT: <time_t>\r", nuts333.c:5025)
# and on the first line of a mail file (c:2469), both shown as they are by .read / .rmail; no other output has digits + CR
_TIME_T_CR = re.compile(rb"(?<![0-9])\d{9,11}\r")
_LONG_DATE = re.compile(rb"\[ [A-Z][a-z]+ \d{1,2} [A-Z][a-z]+ \d{4} at \d\d:\d\d \]")


def normalise(b: bytes) -> bytes:
    b = _SITE_PORT.sub(b"(SITE:PORT)", b)
    b = _LONG_DATE.sub(b"[ DATE ]", b)
    b = _TIME_T_CR.sub(b"T\r", b)
    return _PROMPT.sub(b"<HH:MM, HH:MM, ", b)


def sync_reply(colour: bool, prompt_suffix: bytes = b"") -> bytes:
    """Exactly what ``.version`` makes write_user emit (``nuts333.c:1315-1365``)."""
    if colour:
        return VERSION_LINE + RESET + b"\n\r" + RESET + prompt_suffix
    return VERSION_LINE + b"\n\r" + prompt_suffix


class ScriptError(RuntimeError):
    pass


@dataclass
class Client:
    key: str
    sock: socket.socket
    colour: bool = False
    logged_in: bool = False
    can_sync: bool = True          # False while in a state where .version is not interpreted
    sync_suffix: bytes = b""       # bytes a prompt adds after every command reply (command mode)
    prompt_re: bytes = b""         # regex source for a clock-bearing prompt (prompt on, speech mode)
    hears_broadcasts: bool = True  # False while the user has .ignall on (write_room skips it, nuts333.c:1413)
    buf: bytearray = field(default_factory=bytearray)

    def read_until(self, suffix_or_pred, timeout: float = 10.0) -> bytes:
        """Read until the accumulated buffer satisfies the predicate / ends with the suffix."""
        pred = suffix_or_pred if callable(suffix_or_pred) else (lambda b: b.endswith(suffix_or_pred))
        deadline = time.monotonic() + timeout
        while not pred(bytes(self.buf)):
            left = deadline - time.monotonic()
            if left <= 0:
                raise ScriptError(f"client {self.key}: timed out; have {bytes(self.buf)!r}")
            r, _, _ = select.select([self.sock], [], [], left)
            if r:
                d = self.sock.recv(65536)
                if not d:
                    raise ScriptError(f"client {self.key}: server closed; have {bytes(self.buf)!r}")
                self.buf += d
                # the talker never sets TCP_NODELAY: its second small write to us (the trailing colour
                # reset) waits for our ACK, which the kernel would delay by 40 ms.  Ack at once.
                try:
                    self.sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_QUICKACK, 1)
                except OSError:
                    pass
        out = bytes(self.buf)
        self.buf.clear()
        return out

    def send_raw(self, data: bytes) -> None:
        self.sock.sendall(data)


class Peer:
    """A scripted NUTS-netlink endpoint: plays the *other talker* on a link, byte for byte.

    Either dials the talker's link port (the talker then runs accept_server_connection,
    ``nuts333.c:2892-2942``) or listens and is dialled by a talker booted with
    ``auto_connect YES`` (``nuts333.c:1200-1274``).  Everything it receives is recorded, so the
    fixtures pin the wire protocol itself, not just its effect on telnet clients."""

    def __init__(self, key: str):
        self.key = key
        self.lsock = socket.socket()
        self.lsock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        self.lsock.bind(("127.0.0.1", 0))
        self.lsock.listen(4)
        self.port = self.lsock.getsockname()[1]
        self.sock: socket.socket | None = None
        self.buf = bytearray()

    def dial(self, port: int) -> None:
        self.sock = socket.create_connection(("127.0.0.1", port))
        self.sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)

    def accept(self, timeout: float = 10.0) -> None:
        self.lsock.settimeout(timeout)
        self.sock, _ = self.lsock.accept()
        self.sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)

    def send(self, data: bytes) -> None:
        self.sock.sendall(data)

    def expect(self, suffix: bytes, timeout: float = 10.0) -> bytes:
        deadline = time.monotonic() + timeout
        while not bytes(self.buf).endswith(suffix):
            left = deadline - time.monotonic()
            if left <= 0:
                raise ScriptError(f"peer {self.key}: timed out waiting for {suffix!r}; have {bytes(self.buf)!r}")
            r, _, _ = select.select([self.sock], [], [], left)
            if r:
                d = self.sock.recv(65536)
                if not d:
                    if suffix == b"" or bytes(self.buf).endswith(suffix):
                        break
                    raise ScriptError(f"peer {self.key}: link closed; have {bytes(self.buf)!r}")
                self.buf += d
        out = bytes(self.buf)
        self.buf.clear()
        return out

    def expect_close(self, timeout: float = 10.0) -> bytes:
        """Read until the talker closes the link."""
        deadline = time.monotonic() + timeout
        while True:
            left = deadline - time.monotonic()
            if left <= 0:
                raise ScriptError(f"peer {self.key}: link still open; have {bytes(self.buf)!r}")
            r, _, _ = select.select([self.sock], [], [], left)
            if r:
                d = self.sock.recv(65536)
                if not d:
                    out = bytes(self.buf)
                    self.buf.clear()
                    return out
                self.buf += d

    def close(self) -> None:
        for s in (self.sock, self.lsock):
            try:
                if s is not None:
                    s.close()
            except OSError:
                pass


class Session:
    """Drive several clients against one (or two) talkers and record what each receives."""

    def __init__(self, default_port: int, host: str = "127.0.0.1", talker_ports: list[int] | None = None):
        self.host, self.default_port = host, default_port
        self.talker_ports = talker_ports or [default_port]
        self.wiz_ports: list[int] = []
        self.clients: dict[str, Client] = {}
        self.peers: dict[str, Peer] = {}
        self.steps: list[dict] = []

    # -- netlink peers -----------------------------------------------------------------
    def peer_step(self, key: str, send: bytes, expect: bytes | None, note: str = "", closes: bool = False,
                  client_expect: dict[str, bytes] | None = None) -> None:
        """The scripted peer sends ``send`` on the link and reads until ``expect`` (or until the
        talker closes the link); then every telnet client is synced.  Clients that cannot be
        synced (away over this very link) are read up to the suffix in ``client_expect``."""
        p = self.peers[key]
        if send:
            p.send(send)
        got = p.expect_close() if closes else (p.expect(expect) if expect is not None else b"")
        pre = {k: self.clients[k].read_until(v) for k, v in (client_expect or {}).items()}
        recv = self._collect(None)
        for k, v in pre.items():
            recv[k] = v + recv.get(k, b"")
        recv[key] = got
        what = {"op": "peer", "actor": key, "send": send.decode("latin-1")}
        if note:
            what["note"] = note
        self._record(what, recv)

    def send_only(self, key: str, text: str, peer_expect: dict[str, bytes], note: str = "") -> None:
        """A line whose only effect is on the link (the actor, away on the other talker, gets no
        output of his own): record what the peers receive."""
        self.clients[key].send_raw(text.encode("latin-1") + b"\n")
        got = {k: self.peers[k].expect(v) for k, v in peer_expect.items()}
        recv = self._collect(None)
        recv.update(got)
        what = {"op": "line", "actor": key, "send": text}
        if note:
            what["note"] = note
        self._record(what, recv)

    def peer_expect(self, key: str, suffix: bytes) -> None:
        """Append what the peer has received (up to ``suffix``) to the last recorded step."""
        got = normalise(self.peers[key].expect(suffix)).decode("latin-1")
        self.steps[-1]["recv"][key] = self.steps[-1]["recv"].get(key, "") + got

    # -- plumbing ----------------------------------------------------------------------
    #: the talker's own replies to a toggle of ignall (toggle_ignall, nuts333.c:4463-4476)
    _IGNALL_ON, _IGNALL_OFF = b"You are now ignoring everyone.", b"You will now hear everyone again."
    #: either reply as the talker writes it to the user who toggled: a whole line (write_user turns "\n" into "\n\r", with
    #: ESC[0m before it when colour is on, nuts333.c:1326-1334), at the start of the capture, after a line end, or directly
    #: after the command-mode prompt "COM> " / "COM+> " (nuts333.c:2185-2189), which ends in no newline
    _IGNALL_REPLY = re.compile(rb"(?:\A|[\r\n]|COM\+?> (?:\x1b\[0m)?)(" + re.escape(_IGNALL_ON) + rb"|" + re.escape(_IGNALL_OFF) + rb")(?:\x1b\[0m)?\n")

    def _record(self, what: dict, recv: dict[str, bytes]) -> None:
        # hears_broadcasts follows what the TALKER said, not what was typed: an abbreviated command toggles too,
        # a level-gated refusal does not; the later of the two replies in a capture wins.  Only close()'s wait reads the flag.
        # Only the ACTOR's own capture of the step is read, and only a reply that is a whole line: toggle_ignall
        # answers the user who typed it (nuts333.c:4467,4473), so the same words inside somebody's .say -- the speaker's
        # "You say: ..." echo, a listener's copy -- are payload, not a reply.
        c = self.clients.get(what.get("actor"))
        if c is not None:
            replies = self._IGNALL_REPLY.findall(recv.get(c.key, b""))
            if replies:
                c.hears_broadcasts = replies[-1] == self._IGNALL_OFF
        self.steps.append({**what, "recv": {k: normalise(v).decode("latin-1") for k, v in recv.items() if v}})

    def _sync(self, c: Client) -> bytes:
        """Round trip; returns everything the client had received before the version reply."""
        reply = sync_reply(c.colour, c.sync_suffix)
        c.send_raw(b".version\n")
        if c.prompt_re:
            pat = re.compile(re.escape(sync_reply(c.colour)) + c.prompt_re + rb"\Z", re.S)
            got = c.read_until(lambda b: pat.search(b) is not None)
            return got[: pat.search(got).start()]
        got = c.read_until(reply)
        return got[: -len(reply)]

    def _collect(self, actor: Client | None, first: bytes = b"") -> dict[str, bytes]:
        recv: dict[str, bytes] = {}
        order = ([actor] if actor else []) + [c for c in self.clients.values() if c is not actor]
        # clients that can do the round trip go first: once any of them has its reply the talker
        # (single-threaded) has finished the step, so whatever it wrote to the clients that cannot
        # sync (AFK, away over a link, still logging in) is already queued and a plain drain gets it
        for c in order:
            if c.logged_in and c.can_sync:
                recv[c.key] = (first if c is actor else b"") + self._sync(c)
        for c in order:
            if not (c.logged_in and c.can_sync):
                recv[c.key] = (first if c is actor else b"") + self._drain_nowait(c)
        return {c.key: recv[c.key] for c in order}

    @staticmethod
    def _drain_nowait(c: Client) -> bytes:
        out = bytes(c.buf)
        c.buf.clear()
        while True:
            r, _, _ = select.select([c.sock], [], [], 0)
            if not r:
                return out
            d = c.sock.recv(65536)
            if not d:
                return out
            out += d

    # -- steps -------------------------------------------------------------------------
    def connect(self, key: str, talker: int = 0, wizport: bool = False, expect: bytes = b"Give me a name: ",
                closes: bool = False) -> None:
        port = self.wiz_ports[talker] if wizport else self.talker_ports[talker]
        s = socket.create_connection((self.host, port))
        s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        c = Client(key, s)
        if closes:
            banner = self._read_to_close(c)
            s.close()
        else:
            self.clients[key] = c
            banner = c.read_until(expect)
        self._record({"op": "connect", "actor": key, **({"note": "wizport"} if wizport else {})}, {key: banner})

    @staticmethod
    def _read_to_close(c: Client, timeout: float = 10.0) -> bytes:
        deadline = time.monotonic() + timeout
        while True:
            left = deadline - time.monotonic()
            if left <= 0:
                raise ScriptError(f"client {c.key}: server did not close; have {bytes(c.buf)!r}")
            r, _, _ = select.select([c.sock], [], [], left)
            if r:
                try:
                    d = c.sock.recv(65536)
                except ConnectionResetError:
                    d = b""
                if not d:
                    out = bytes(c.buf)
                    c.buf.clear()
                    return out
                c.buf += d

    def dialog(self, key: str, send: str, expect: bytes | None = None, closes: bool = False, note: str = "",
               logged_in: bool | None = None, **flags) -> None:
        """Pre-login conversation: send one line, read up to ``expect`` (or to the close)."""
        c = self.clients[key]
        c.send_raw(send.encode("latin-1") + b"\n")
        if closes:
            mine = self._read_to_close(c)
            c.sock.close()
            del self.clients[key]
        else:
            mine = c.read_until(expect)
        if logged_in is not None and not closes:
            c.logged_in = logged_in
        if flags:
            self.set_flags(key, **flags)
        recv = self._collect(None) if not closes or self.clients else {}
        recv[key] = mine + recv.get(key, b"")
        what = {"op": "dialog", "actor": key, "send": send}
        if note:
            what["note"] = note
        self._record(what, recv)

    def login(self, key: str, name: str, password: str = "test", colour: bool = False,
              sync_suffix: bytes = b"", prompt_re: bytes = b"") -> None:
        c = self.clients[key]
        c.send_raw(name.encode() + b"\n")
        # the account's colour flag is live as soon as the name is accepted (load_user_details,
        # nuts333.c:1510,1622), so the password prompt and the IAC WILL ECHO that follows it
        # (echo_off, nuts333.c:1814-1822) are each followed by a reset when colour is on
        a = c.read_until(b"\xff\xfb\x01" + (RESET if colour else b""))
        c.send_raw(password.encode() + b"\n")
        # login ends with look()'s topic line and, in command mode, the COM> prompt
        c.colour = colour
        c.sync_suffix = sync_suffix
        c.prompt_re = prompt_re
        tail = (b"has been set yet." + (RESET if colour else b"") + b"\n\r" + (RESET if colour else b""))
        if prompt_re:
            pat = re.compile(re.escape(tail) + prompt_re + rb"\Z", re.S)
            b = c.read_until(lambda x: pat.search(x) is not None)
        else:
            b = c.read_until(tail + sync_suffix)
        c.logged_in = True
        recv = self._collect(None)
        recv[key] = a + b + recv.get(key, b"")
        self._record({"op": "login", "actor": key, "name": name}, recv)

    def line(self, key: str, text: str, note: str = "", expect: bytes | None = None, **flags) -> None:
        """Send one input line, then sync everybody.  ``flags`` (see :meth:`set_flags`) describe
        what the line does to the actor's own output state (``.colour``, ``.mode``, ``.prompt``)
        and take effect before the sync.  With ``expect`` the actor is not synced (the sync
        command would overwrite the "." repeat buffer, nuts333.c:166-174): its capture ends at
        the given suffix instead."""
        c = self.clients[key]
        c.send_raw(text.encode("latin-1") + b"\n")
        if flags:
            self.set_flags(key, **flags)
        if expect is not None:
            mine = c.read_until(expect)
            was, c.can_sync = c.can_sync, False
            try:
                recv = self._collect(None)
            finally:
                c.can_sync = was
            recv[key] = mine + recv.get(key, b"")
            self._record({"op": "line", "actor": key, "send": text, **({"note": note} if note else {})}, recv)
            return
        # The sync command must not share a read() with this line (the talker would drop it), so the
        # talker has to consume the line first.  Not every line produces output for its sender (an
        # emote by a user who ignores everyone, a clone told to hear nothing), so instead of waiting
        # for output: wait until the line sits in the talker's receive queue (our send queue is
        # acknowledged empty), then let ANOTHER client do two round trips -- the talker serves every
        # ready socket per select() pass, so after the second reply the pass that held our line is over.
        helper = next((h for h in self.clients.values() if h is not c and h.logged_in and h.can_sync), None)
        if helper is None or not (c.logged_in and c.can_sync):
            first = c.read_until(lambda b: len(b) > 0)
            recv = self._collect(c, first)
        else:
            self._await_acked(c)
            early = self._sync(helper) + self._sync(helper)
            recv = self._collect(c)
            recv[helper.key] = early + recv.get(helper.key, b"")
        what = {"op": "line", "actor": key, "send": text}
        if note:
            what["note"] = note
        self._record(what, recv)

    def raw(self, key: str, chunks: Iterable[bytes], note: str = "", expect_output: bool = True) -> None:
        """Send raw byte chunks (one TCP segment each, in order) -- for input-framing cases."""
        c = self.clients[key]
        chunks = list(chunks)
        first = b""
        for i, ch in enumerate(chunks):
            c.send_raw(ch)
            if i + 1 < len(chunks):
                # let the talker consume this segment on its own before the next is sent
                self._settle(c)
        helper = next((h for h in self.clients.values() if h is not c and h.logged_in and h.can_sync), None)
        if helper is not None and c.logged_in and c.can_sync:
            # as in line(): make sure the talker has consumed the last segment before the sync command is sent
            # (echoed characters of an earlier segment must not be mistaken for the reply to the last one)
            self._await_acked(c)
            early = self._sync(helper) + self._sync(helper)
            recv = self._collect(c)
            recv[helper.key] = early + recv.get(helper.key, b"")
        else:
            if expect_output:
                first = c.read_until(lambda b: len(b) > 0)
            recv = self._collect(c, first)
        self._record({"op": "raw", "actor": key, "send": [ch.decode("latin-1") for ch in chunks], "note": note}, recv)

    @staticmethod
    def _await_acked(c: Client, timeout: float = 5.0) -> None:
        """Block until everything we sent has been acknowledged by the peer's TCP stack."""
        import fcntl, struct, termios
        deadline = time.monotonic() + timeout
        while time.monotonic() < deadline:
            if struct.unpack("i", fcntl.ioctl(c.sock.fileno(), termios.TIOCOUTQ, b"\0\0\0\0"))[0] == 0:
                return
            time.sleep(0.0002)
        raise ScriptError(f"client {c.key}: data never acknowledged")

    def raw_dialog(self, key: str, chunks: Iterable[bytes], expect: bytes, note: str = "", logged_in: bool | None = None) -> None:
        """Pre-login conversation typed in pieces (a character-mode client): send the chunks one segment at
        a time, then read up to ``expect``."""
        c = self.clients[key]
        chunks = list(chunks)
        for i, ch in enumerate(chunks):
            c.send_raw(ch)
            if i + 1 < len(chunks):
                self._settle(c)
        mine = c.read_until(expect)
        if logged_in is not None:
            c.logged_in = logged_in
        recv = self._collect(None)
        recv[key] = mine + recv.get(key, b"")
        self._record({"op": "raw", "actor": key, "send": [ch.decode("latin-1") for ch in chunks], "note": note}, recv)

    def _settle(self, c: Client, quiet: float = 0.05) -> None:
        """Wait until the talker has read the bytes we just sent (its socket receive queue is
        empty), without sending anything ourselves.  Uses SIOCOUTQ on our side: unacked+unsent
        bytes drop to zero once the peer's stack has them; then give select() one tick."""
        import fcntl, struct, termios
        deadline = time.monotonic() + 2.0
        while time.monotonic() < deadline:
            outq = struct.unpack("i", fcntl.ioctl(c.sock.fileno(), termios.TIOCOUTQ, b"\0\0\0\0"))[0]
            if outq == 0:
                break
            time.sleep(0.001)
        time.sleep(quiet)

    def set_flags(self, key: str, *, colour: bool | None = None, can_sync: bool | None = None,
                  sync_suffix: bytes | None = None, prompt_re: bytes | None = None, hears_broadcasts: bool | None = None) -> None:
        c = self.clients[key]
        if hears_broadcasts is not None:
            c.hears_broadcasts = hears_broadcasts
        if colour is not None:
            c.colour = colour
        if can_sync is not None:
            c.can_sync = can_sync
        if sync_suffix is not None:
            c.sync_suffix = sync_suffix
        if prompt_re is not None:
            c.prompt_re = prompt_re

    def close(self, key: str) -> None:
        """Leave by closing the socket (never .quit); the others see the SIGN OFF broadcast."""
        c = self.clients.pop(key)
        others = list(self.clients.values())
        c.sock.close()
        # The talker notices on its next select() and writes the SIGN OFF broadcast to every listener in one pass
        # (single-threaded).  Wait PASSIVELY for the first byte of it on any other client -- no round trips yet, so
        # that the talker's own sequence of write(2) calls is a function of the script alone (the write-order parity
        # test relies on that) -- then one round of syncs collects it everywhere.  Nobody listening (a half-open login
        # closed, everyone ignoring): the short wait (0.5 s, passive) runs out and the round of syncs records the silence.
        # Only listeners the broadcast can reach are waited on: logged in (a half-open login dropped by the talker
        # reads EOF for ever) and not ignoring everything (write_room skips ignall users, nuts333.c:1413; the editor sets the
        # same skip but no script closes a peer while another is in it).  Listeners on the OTHER talker of a linked pair stay
        # in: write_room(NULL, ...) at nuts333.c:1782 also reaches REMOTE_TYPE users, whose copy travels over the link.
        listeners = [o.sock for o in others if o.logged_in and o.hears_broadcasts]
        # Content-aware, still passive: when the leaver was logged in, look (MSG_PEEK -- nothing is consumed, nothing
        # is sent) for the broadcast itself for up to 5 s.  A talker descheduled for more than half a second on a busy
        # host, or an unrelated byte from a netlink relay, must not end the wait early and push SIGN OFF into the next
        # step's capture.  A half-open login leaves silently (nuts333.c:1770-1775): the short wait is enough there, and so
        # it is when nobody is left who could be sent the broadcast.
        expect_broadcast = bool(listeners) and c.logged_in
        deadline = time.monotonic() + (5.0 if expect_broadcast else 0.5)
        seen = False
        if not listeners and others and c.logged_in:
            # a logged-in leaver whom nobody can be sent: the short wait, passive, nothing to look for.  A half-open login
            # leaves without a word to anyone (nuts333.c:1770-1775) and nobody is waited on: no sleep at all
            time.sleep(max(0.0, deadline - time.monotonic()))
        while listeners and not seen and time.monotonic() < deadline:
            ready, _, _ = select.select(listeners, [], [], max(0.0, deadline - time.monotonic()))
            if not expect_broadcast:
                break
            for sock in ready:
                try:
                    peeked = sock.recv(65536, socket.MSG_PEEK)
                except OSError:
                    peeked = b""
                if b"SIGN OFF:" in peeked:
                    seen = True
                    break
                if not peeked:          # EOF (or a dead socket) stays select()-ready for ever: stop waiting on it
                    listeners.remove(sock)
            if not listeners and not seen:          # nobody left to hear it: what remains is the short, silent wait
                time.sleep(max(0.0, min(0.5, deadline - time.monotonic())))
            elif ready and not seen:
                time.sleep(0.005)       # bytes that are not the broadcast are pending: do not spin on them
        recv: dict[str, bytes] = self._collect(None) if others else {}
        self._record({"op": "close", "actor": key}, recv)

    def shutdown(self) -> None:
        for c in list(self.clients.values()):
            try:
                c.sock.close()
            except OSError:
                pass
        self.clients.clear()
        for p in self.peers.values():
            p.close()
        self.peers.clear()
