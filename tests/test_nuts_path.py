"""Unit parity for the pure functions of the restatement (oracle/nuts_path.c), via ctypes.

Expected values come from the reference, not from reading our own code back:
* tests/golden/vectors/transducer.json -- 314 seeded lines said through the real reference
  build, with the bytes a colour-on and a colour-off listener received
  (tests/golden/make_vectors.py);
* the known answers below for framing / dispatch are the behaviours visible in the golden
  transcripts (tests/golden/framing.json, speech_colour_off.json, errors.json).
"""
from __future__ import annotations

BOTH_TIERS = True       # tests/conftest.py: every test here also runs in the gpu tier, on the MI355X box's host

import ctypes
import json
import re
from pathlib import Path

import pytest
from hypothesis import given, settings, strategies as st

REPO = Path(__file__).resolve().parent.parent
VECTORS = json.loads((REPO / "tests" / "golden" / "vectors" / "transducer.json").read_text())
ANSI = re.compile(rb"\x1b\[\d+m")


@pytest.fixture(scope="module")
def lib(built):
    lib = ctypes.CDLL(str(REPO / "oracle" / "_build" / "libnuts_path.so"))
    lib.np_transduce.restype = ctypes.c_size_t
    lib.np_transduce.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
    lib.np_write_count.argtypes = [ctypes.c_char_p, ctypes.c_int]
    lib.np_colour_com_strip.restype = ctypes.c_size_t
    lib.np_colour_com_strip.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
    lib.np_terminate.argtypes = [ctypes.c_char_p]
    lib.np_wordfind.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    lib.np_remove_first.restype = ctypes.c_char_p
    lib.np_remove_first.argtypes = [ctypes.c_char_p]
    lib.np_command_lookup.argtypes = [ctypes.c_char_p]
    lib.np_command_name.restype = ctypes.c_char_p
    lib.np_say_verb.restype = ctypes.c_char_p
    lib.np_say_verb.argtypes = [ctypes.c_char_p]
    lib.np_contains_swearing.argtypes = [ctypes.c_char_p]
    lib.np_record.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_char_p]
    return lib


def transduce(lib, s: bytes, colour: int) -> bytes:
    out = ctypes.create_string_buffer(len(s) * 6 + 64)
    n = lib.np_transduce(s, colour, out, len(out))
    assert n <= len(out)
    return out.raw[:n]


# ---------------------------------------------------------------- transducer vs the reference
def test_vector_file_is_what_the_generator_promises():
    assert VECTORS["seed"] == 333 and len(VECTORS["vectors"]) == 314


@pytest.mark.parametrize("colour,key", [(1, "colour_on"), (0, "colour_off")])
def test_transducer_matches_reference_vectors(lib, colour, key):
    bad = []
    for v in VECTORS["vectors"]:
        src = ("Bobby says: " + v["line"] + "\n").encode("latin-1")
        if transduce(lib, src, colour) != v[key].encode("latin-1"):
            bad.append(v["line"])
    assert not bad, f"{len(bad)} of {len(VECTORS['vectors'])} differ, first: {bad[0]!r}"


def test_transducer_matches_reference_for_the_speaker_too(lib):
    for v in VECTORS["vectors"]:
        src = ("You say: " + v["line"] + "\n").encode("latin-1")
        assert transduce(lib, src, 0) == v["self"].encode("latin-1"), v["line"]


def test_known_answers_from_golden_markup(lib):
    # tests/golden/markup.json, colour-off and colour-on listener
    assert transduce(lib, b"escaped /~FR stays text, /~ alone, // and / ~\n", 0) == b"escaped ~FR stays text, ~ alone, // and / ~\n\r"
    assert transduce(lib, b"adjacent ~FR~BGcodes~RS~RS\n", 1) == b"adjacent \x1b[31m\x1b[42mcodes\x1b[0m\x1b[0m\x1b[0m\n\r\x1b[0m"
    assert transduce(lib, b"~~ double tilde ~~FR and /~~FG\n", 0) == b"~~ double tilde ~ and ~\n\r"
    assert transduce(lib, b"\xff\xfb\x01", 0) == b"\xff\xfb\x01"          # echo_off bytes pass through


def test_write_boundaries(lib):
    """One write per <=1000 staged bytes, plus one for the trailing reset when colour is on
    (nuts333.c:1359-1365): 67-byte broadcast = 1 write (2 with colour)."""
    line = b"Uaaa says: " + b"x" * 54 + b"\n"
    assert len(transduce(lib, line, 0)) == 67
    assert lib.np_write_count(line, 0) == 1 and lib.np_write_count(line, 1) == 2
    assert lib.np_write_count(b"a" * 999 + b"\n", 0) == 2      # newline needs 6 spare bytes: early flush
    assert lib.np_write_count(b"a" * 2500, 0) == 3
    assert lib.np_write_count(b"", 0) == 0 and lib.np_write_count(b"", 1) == 1


def test_stage_keeps_its_fill_level_across_strings(lib):
    """np_stage_*: the staging buffer as more() uses it (nuts333.c:2250-2296) -- fed line by line, flushed once at the end.
    The chunk sizes are the reference's own write(2) sizes for this file, logged from the real talker by
    tests/test_harness.py::test_more_flushes_where_the_reference_does_for_long_banners: 995 | 995 | 1000 | rest."""
    import scenarios

    class Stage(ctypes.Structure):
        _fields_ = [("buff", ctypes.c_char * 1008), ("pos", ctypes.c_int)]

    EMIT = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.POINTER(ctypes.c_char), ctypes.c_size_t)
    chunks = []
    emit = EMIT(lambda ctx, buf, n: chunks.append(ctypes.string_at(buf, n)))
    lib.np_stage_init.argtypes = [ctypes.POINTER(Stage)]
    lib.np_stage_feed.argtypes = [ctypes.POINTER(Stage), ctypes.c_char_p, ctypes.c_int, EMIT, ctypes.c_void_p]
    lib.np_stage_flush.argtypes = [ctypes.POINTER(Stage), EMIT, ctypes.c_void_p]
    st_ = Stage()
    lib.np_stage_init(ctypes.byref(st_))
    for line in scenarios.LONG_MOTD1.splitlines(keepends=True):
        lib.np_stage_feed(ctypes.byref(st_), line.encode(), 0, emit, None)
    lib.np_stage_flush(ctypes.byref(st_), emit, None)
    assert [len(c) for c in chunks][:3] == [995, 995, 1000]
    whole = b"".join(chunks)
    assert whole == b"".join(transduce(lib, l.encode(), 0) for l in scenarios.LONG_MOTD1.splitlines(keepends=True))
    assert whole.endswith(b"~FR escaped, ~ZZ unknown, the end of motd1\n\r\n\r")


PRINTABLE = st.text(alphabet=st.sampled_from(list("abcXYZ ~/FRSOLBG0123\n")), max_size=300)


@settings(max_examples=300, deadline=None)
@given(PRINTABLE)
def test_colour_off_equals_colour_on_minus_ansi(lib, s):
    b = s.encode()
    assert ANSI.sub(b"", transduce(lib, b, 1)) == transduce(lib, b, 0)


@settings(max_examples=300, deadline=None)
@given(PRINTABLE)
def test_newline_accounting(lib, s):
    b = s.encode()
    out = transduce(lib, b, 0)
    assert out.count(b"\n\r") == b.count(b"\n")
    assert len(out) <= len(b) + b.count(b"\n")


def test_colour_com_strip(lib):
    out = ctypes.create_string_buffer(256)
    lib.np_colour_com_strip(b"~OLbold~RS /~FR ~ZZ ~", out, 256)
    assert out.value == b"bold / ~ZZ ~"      # unlike write_user, strip() does not know the /~ escape (c:2588-2610)


# ---------------------------------------------------------------- framing
def words(lib, s: bytes):
    buf = ctypes.create_string_buffer(10 * 41)
    n = lib.np_wordfind(s, buf)
    return n, [buf.raw[i * 41:(i + 1) * 41].split(b"\0")[0] for i in range(10)]


def test_terminate(lib):
    for raw, want in [(b"pipeA\npipeB\n", b"pipeA"), (b"crlf line\r\n", b"crlf line"), (b"tab\there\n", b"tab"),
                      (b"high bit \xe9\xe8 cut\n", b"high bit "), (b"\n", b"")]:
        buf = ctypes.create_string_buffer(raw + b"\0" * 8, 1100)
        assert lib.np_terminate(buf) == len(want) and buf.value == want
    buf = ctypes.create_string_buffer(b"x" * 1050, 1100)
    assert lib.np_terminate(buf) == 999 and len(buf.value) == 999


def test_wordfind(lib):
    n, w = words(lib, b"  .tell   bobby hello there ")
    assert n == 4 and w[:4] == [b".tell", b"bobby", b"hello", b"there"]
    n, w = words(lib, b"w1 w2 w3 w4 w5 w6 w7 w8 w9 w10 w11 w12")
    assert n == 9 and w[9] == b"w10"                               # ten slots filled reports nine (c:430-431)
    n, w = words(lib, b".tell " + b"b" * 50 + b" x")
    assert n == 4 and w[1] == b"b" * 39 and w[2] == b"b" * 11      # long word spills into the next slot
    assert words(lib, b"")[0] == 0 and words(lib, b"   ")[0] == 0


def test_remove_first(lib):
    assert lib.np_remove_first(b".tell bobby hi") == b"bobby hi"
    assert lib.np_remove_first(b"   .say   x  y") == b"x  y"
    assert lib.np_remove_first(b"single") == b""


# ---------------------------------------------------------------- dispatch
def test_command_table(lib):
    assert lib.np_command_count() == 92
    name = lambda s: lib.np_command_name(lib.np_command_lookup(s))
    assert name(b"s") == b"say" and name(b"sh") == b"shout" and name(b"se") == b"semote"
    assert name(b"t") == b"tell" and name(b"rev") == b"review" and name(b"revt") == b"revtell"
    assert name(b"i") == b"ignall" and name(b"igns") == b"ignshout" and name(b"c") == b"connect"
    assert lib.np_command_lookup(b"bogus") == -1
    lvl = lambda s: lib.np_command_level(lib.np_command_lookup(s))
    assert lvl(b"say") == 0 and lvl(b"shout") == 1 and lvl(b"tell") == 1 and lvl(b"invis") == 3 and lvl(b"shutdown") == 4


def test_say_verb_and_swear_filter(lib):
    assert lib.np_say_verb(b"hello") == b"say" and lib.np_say_verb(b"really?") == b"ask" and lib.np_say_verb(b"no!") == b"exclaim"
    assert lib.np_contains_swearing(b"what the FuCk") == 1 and lib.np_contains_swearing(b"scunthorpe") == 1
    assert lib.np_contains_swearing(b"clean line") == 0


# ---------------------------------------------------------------- fan-out predicate
class Listener(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("login", "has_room", "same_room", "ignall", "ignshout", "is_sender")]


def test_fanout_predicate_truth_table(lib):
    SHOUT, SEMOTE, SAY = 4, 7, 3
    admit = lambda rm_null=0, force=0, com=SAY, **kw: lib.np_fanout_admits(
        ctypes.byref(Listener(**{"login": 0, "has_room": 1, "same_room": 1, "ignall": 0, "ignshout": 0, "is_sender": 0, **kw})),
        rm_null, force, com)
    assert admit() == 1
    assert admit(login=3) == 0 and admit(has_room=0) == 0 and admit(is_sender=1) == 0
    assert admit(same_room=0) == 0 and admit(same_room=0, rm_null=1) == 1
    assert admit(ignall=1) == 0 and admit(ignall=1, force=1) == 1
    assert admit(ignshout=1, com=SHOUT) == 0 and admit(ignshout=1, com=SEMOTE) == 0 and admit(ignshout=1, com=SAY) == 1


# ---------------------------------------------------------------- review ring
def test_record_ring(lib):
    ring = ctypes.create_string_buffer(5 * 202)
    rev = ctypes.c_int(0)
    for i in range(7):
        lib.np_record(ring, 5, ctypes.byref(rev), f"line {i}\n".encode())
    assert rev.value == 2
    slot = lambda i: ring.raw[i * 202:(i + 1) * 202].split(b"\0")[0]
    assert slot(0) == b"line 5\n" and slot(1) == b"line 6\n" and slot(2) == b"line 2\n"
    lib.np_record(ring, 5, ctypes.byref(rev), b"y" * 250 + b"\n")
    assert slot(2) == b"y" * 200 + b"\n"                          # cut at 200, newline forced (c:2066-2068)


# ---------------------------------------------------------------- ABI surface
def test_library_exports_every_declared_symbol(lib):
    """oracle/nuts_path.h is the only C header in the repo; the .so must export all of it."""
    header = (REPO / "oracle" / "nuts_path.h").read_text()
    declared = sorted(set(re.findall(r"\b(np_[a-z_]+)\s*\(", header)) - {"np_emit_fn"})
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in nuts_path.h but not exported"
