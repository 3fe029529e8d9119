"""Transcript scenarios: the scripted inputs whose byte-exact outputs pin the hot path.

Shared by ``tests/golden/make_golden.py`` (which runs them against the reference build,
``oracle/_ref/nuts333``, and writes ``tests/golden/*.json``) and by the parity tests (which
replay them against ``oracle/_build/talker_port`` and, where it is present, the reference
again).  Each scenario returns ``(TalkerConfig kwargs, accounts, script)`` where ``script``
is a callable driving a :class:`nuts333_amd.transcript.Session`.

What they pin, by reference function:
  speech_*      say / shout / tell + siblings, write_room_except, write_user   c:4062-4300, 1401-1429, 1291-1366
  markup        the ``~XX`` / ``/~`` transducer in both colour states           c:1315-1362
  filters       ignall, ignshout, igntell, invisible sender, room scoping       c:1410-1415, 4156-4167, 4096
  errors        usage / level / muzzle / unknown-user early-outs                c:3782-3784, 4068-4079, 4110-4118, 4135-4148
  framing       terminate, char-mode assembly, pipelining drop, "." repeat      c:136-175, 369-411
  review        record / record_tell ring buffers                               c:2062-2082
  prompt        prompt() in speech and command mode                             c:2174-2197
  afk_bcast     the AFK branch of the main loop (incl. locked sessions), bcast (force_listen),  c:180-203, 4772-4788, 6527-6565,
                wizshout (write_level), cls                                              7409-7454, 2636-2642
  charecho      server-side echo for character-mode clients                                      c:369-399, 6881-6893
  clones        the clone branch of write_room_except: relay to the owner, hear all/swears/      c:1416-1426, 4085-4090, 7100-7357,
                nothing, csay, switch, destroy, clean-up at logout                        2870-2882
  rooms         go / move_user / look: adjacency, prefix names, teleport, private     c:3942-4004, 4305-4459, 2412-2421
                rooms, invisible movement
  login_paths   accept + 3-stage login incl. every error exit, new account, wizport,   c:263-311, 1451-1606, 1645-1673
                ban list, hung-login takeover; the .D record written at logout
  capacity      max_users on the main port only                                        c:287-291
  long_motd     more(): banners over 1 KB through the 1000-byte staging buffer            c:2205-2296
  netlink       two talkers: TRANS/GRANTED, ACT relay, MSG..EMSG frames, PRM,   c:2946-3073, 3077-3285, 1299-1306,
                REMVD on the way home, offsite tell, home execution               3452-3479, 3787-3806, 4168-4172
  netlink_wire_* a scripted peer speaks the link protocol itself: every verb's    c:2892-2942, 2946-3073, 3077-3479,
                bytes in both directions, split/merged segments, denials, link     3689-3746, 4305-4375
                shutdown; _legacy: what peers announcing 3.3.2 / 3.1.0 are sent    c:1299-1300, 3093-3097, 3129-3146
"""
from __future__ import annotations

from nuts333_amd import provision as pv

A, B, C, D = "Alice", "Bobby", "Carol", "Dave"


def mask_user_record(text: str) -> str:
    """A saved .D record with its four wall-clock fields masked (DOCS/userdata_format:7-15)."""
    lines = text.split("\n")
    if len(lines) > 1:
        f = lines[1].split(" ")
        lines[1] = " ".join(["T"] * 4 + f[4:])
    return "\n".join(lines)


_PT = __import__("re").compile(r"PT: \d+\r")
_LONG_DATE = __import__("re").compile(r"\[ [A-Z][a-z]+ \d{1,2} [A-Z][a-z]+ \d{4} at \d\d:\d\d \]")


def mask_file(rel: str, text: str) -> str:
    """On-disk side effects with their wall-clock fields masked, by file type:
    ``.D`` user record (four timestamps on line 2, DOCS/userdata_format:7-15);
    ``.B`` board: ``PT: <time_t>\\r`` in front of each header and the date stamp in it (nuts333.c:5020-5028);
    ``.M`` mail: ``<time_t>\\r`` opening the file (nuts333.c:2468-2469) and the date stamp of each header (c:2493-2496)."""
    if rel.endswith(".D"):
        return mask_user_record(text)
    text = _LONG_DATE.sub("[ DATE ]", text)
    if rel.endswith(".B"):
        return _PT.sub("PT: T\r", text)
    if rel.endswith(".M"):
        head, sep, rest = text.partition("\r")
        return ("T" if head.isdigit() else head) + sep + rest
    return text


def _acc(name, **kw):
    return pv.Account(name, **{"desc": f"is {name.lower()}", **kw})


def speech_colour_off():
    accounts = [_acc(A), _acc(B), _acc(C)]

    def script(s):
        for k, n in (("a", A), ("b", B), ("c", C)):
            s.connect(k); s.login(k, n)
        s.line("c", ".go hallway")                    # c leaves the room: say must not reach it, shout must
        s.line("a", "hello world")
        s.line("a", "is anybody there?")
        s.line("a", "watch out!")
        s.line("b", ".say explicit say command")
        s.line("a", ".shout hi all")
        s.line("a", "! shortcut shout")
        s.line("a", ".tell bobby a private word")
        s.line("a", "> carol a question for you?")
        s.line("b", ".tell al partial-name match fails; exact-then-substring lookup")
        s.line("b", ".tell lic substring of Alice")
        s.line("a", ";waves")
        s.line("a", ".emote nods slowly")
        s.line("a", "#cheers loudly")
        s.line("a", ".semote grins")
        s.line("a", "< bobby winks")
        s.line("a", ".pemote carol bows")
        s.line("a", "- an echoed line")
        s.line("a", ".sh abbreviated command name")
        s.line("a", ".s first match in the table wins")   # 'say' precedes 'shout' (nuts333.h:157-159)
        s.close("b")
        s.line("a", "after bobby left")

    return {}, accounts, script


def speech_colour_mixed():
    accounts = [_acc(A, colour=1), _acc(B, colour=0), _acc(C, colour=1)]

    def script(s):
        s.connect("a"); s.login("a", A, colour=True)
        s.connect("b"); s.login("b", B)
        s.connect("c"); s.login("c", C, colour=True)
        s.line("a", "plain say, colour recipients get resets")
        s.line("b", ".shout bold prefix for colour users")
        s.line("b", ".tell alice bold tell")
        s.line("c", ".tell bobby stripped for bobby")
        s.line("a", ";emotes")
        s.line("b", "#semotes")
        s.line("b", ".pemote alice private emote to a colour user")
        s.line("a", "< bobby private emote to a plain user")
        s.line("b", "- an echo reaches colour users with resets")
        s.line("c", ".echo ~FGgreen~RS echo, stripped for bobby")
        s.line("b", ".colour", colour=True)
        s.line("a", "now bobby has colour too")
        s.line("a", ".colour", colour=False)
        s.line("c", ".shout alice switched it off")

    return {}, accounts, script


def markup():
    accounts = [_acc(A, colour=1), _acc(B, colour=0)]

    def script(s):
        s.connect("a"); s.login("a", A, colour=True)
        s.connect("b"); s.login("b", B)
        codes = "RS OL UL LI RV FK FR FG FY FB FM FT FW BK BR BG BY BB BM BT BW".split()
        s.line("b", "all codes " + " ".join(f"~{c}{c.lower()}" for c in codes))
        s.line("b", "unknown ~ZZ code and ~fr lower case")
        s.line("b", "escaped /~FR stays text, /~ alone, // and / ~")
        s.line("b", "trailing tilde ~")
        s.line("b", "one letter after ~F")
        s.line("b", "adjacent ~FR~BGcodes~RS~RS")
        s.line("b", "~OLstarts with a code")
        s.line("b", ".shout ~FYyellow shout~RS done")
        s.line("b", ".tell alice ~ULunderlined~RS tell /~OL")
        s.line("a", "~~ double tilde ~~FR and /~~FG")
        s.line("a", "x" * 40 + " ~FR" + "y" * 40)

    return {}, accounts, script


def filters():
    accounts = [_acc(A), _acc(B), _acc(C), _acc(D, level=3)]

    def script(s):
        for k, n in (("a", A), ("b", B), ("c", C), ("d", D)):
            s.connect(k); s.login(k, n)
        s.line("b", ".ignshout")
        s.line("a", ".shout bobby should not hear this")
        s.line("a", "#nor this semote")
        s.line("a", "but bobby hears a say")
        s.line("b", ".ignshout")
        s.line("c", ".ignall")
        s.line("a", "carol ignores everything")
        s.line("a", ".shout including shouts")
        s.line("a", ".tell carol and tells")
        s.line("d", ".tell carol but an ARCH gets through")
        s.line("c", ".ignall")
        s.line("b", ".igntell")
        s.line("a", ".tell bobby ignored tell")
        s.line("a", "< bobby ignored pemote")
        s.line("b", ".igntell")
        s.line("d", ".invis")
        s.line("d", "a presence speaks")
        s.line("d", ".shout a presence shouts")
        s.line("d", ".tell alice a presence tells")
        s.line("d", ";a presence emotes")
        s.line("d", ".vis")
        s.line("d", "visible again")

    return {}, accounts, script


def errors():
    accounts = [_acc(A), _acc(B, muzzled=2), _acc(C, level=0), _acc(D, command_mode=1)]

    def script(s):
        s.connect("a"); s.login("a", A)
        s.connect("b"); s.login("b", B)
        s.connect("c"); s.login("c", C)
        s.connect("d"); s.login("d", D, sync_suffix=b"COM> ")
        s.line("a", ".shout")
        s.line("a", ".tell")
        s.line("a", ".tell bobby")
        s.line("a", ".tell nobody hello there")
        s.line("a", ".tell alice hello me")
        s.line("a", ".say")
        s.line("a", ".bogus command")
        s.line("a", ". ")
        s.line("a", ".pemote alice self")
        s.line("a", ".pemote")
        s.line("a", ".echo")
        s.line("a", ";")
        s.line("a", "#")
        s.line("b", "muzzled say")
        s.line("b", ".shout muzzled shout")
        s.line("b", ".tell alice muzzled tell")
        s.line("b", ";muzzled emote")
        s.line("b", "#muzzled semote")
        s.line("b", "- muzzled echo")
        s.line("c", "a NEW user may say")
        s.line("c", ".shout but not shout")
        s.line("c", ".tell alice nor tell")
        s.line("d", "say")                            # command mode: bare word is a command -> "Say what?"
        s.line("d", "say in command mode")
        s.line("d", "shout from command mode")
        s.line("d", "hello")                          # unknown command in command mode

    return {}, accounts, script


def swearing():
    accounts = [_acc(A), _acc(B)]

    def script(s):
        s.connect("a"); s.login("a", A)
        s.connect("b"); s.login("b", B)
        s.line("a", "what the FuCk")
        s.line("a", ".shout oh shit")
        s.line("a", ";says cunt")
        s.line("a", "scunthorpe problem")
        s.line("a", ".tell bobby shit is allowed in tells")
        s.line("a", "#shit is allowed in semotes")
        s.line("a", "clean line")

    return {"ban_swearing": True}, accounts, script


def framing():
    accounts = [_acc(A), _acc(B)]

    def script(s):
        s.connect("a"); s.login("a", A)
        s.connect("b"); s.login("b", B)
        s.raw("a", [b"pipeA\npipeB\npipeC\n"], note="pipelined lines in one segment: only the first survives")
        s.raw("a", [b"crlf line\r\n"], note="telnet CR LF: cut at CR")
        s.raw("a", [b"tab\there\n"], note="cut at the first control character")
        s.raw("a", [b"char", b"mode ", b"typing\n"], note="character-mode client: bytes without newline are buffered")
        s.raw("a", [b"del", b"ete\x7f\x7f\x7fay\n"], note="DEL erases buffered characters")
        s.raw("a", [b"ab", b"\x08\x08\x08\x08cd\n"], note="backspace below zero is ignored")
        s.raw("a", [b"\xff\xfd\x01", b"after IAC\n"], note="IAC replies are ignored", expect_output=True)
        s.line("a", "repeat me", expect=b"You say: repeat me\n\r")
        s.line("a", ".", expect=b"You say: repeat me\n\r", note="a lone dot re-runs the previous line")
        s.line("a", ".", expect=b"You say: repeat me\n\r", note="and again: the dot itself is not stored")
        s.line("a", "   leading spaces kept")
        s.line("a", "w1 w2 w3 w4 w5 w6 w7 w8 w9 w10 w11 w12")
        s.line("a", "x" * 300)
        s.line("a", ".tell " + "b" * 50 + " long first word")
        s.line("a", ".shout " + "long " * 190)
        s.line("a", "high bit bytes \xe9\xe8 are cut")

    return {}, accounts, script


def review():
    accounts = [_acc(A), _acc(B)]

    def script(s):
        s.connect("a"); s.login("a", A)
        s.connect("b"); s.login("b", B)
        for i in range(17):
            s.line("a" if i % 2 == 0 else "b", f"review line {i:02d}")
        s.line("a", ";emotes into the buffer")
        s.line("a", "- echoes into the buffer")
        s.line("a", ".shout shouts are not recorded")
        s.line("a", ".review")
        for i in range(7):
            s.line("a", f".tell bobby tell number {i}")
        s.line("a", "< bobby pemote is recorded too")
        s.line("b", ".revtell")
        s.line("a", ".revtell")
        s.line("a", "y" * 250)
        s.line("b", ".review")

    return {}, accounts, script


def prompts():
    accounts = [_acc(A, prompt=1), _acc(B, command_mode=1, level=3), _acc(C, prompt=1, colour=1, level=3)]
    pa = rb"<\d\d:\d\d, \d\d:\d\d, Alice>\n\r"
    pc = rb"\x1b\[36m<\d\d:\d\d, \d\d:\d\d, Carol>\x1b\[0m\n\r\x1b\[0m"
    pc_invis = rb"\x1b\[36m<\d\d:\d\d, \d\d:\d\d, Carol\+>\x1b\[0m\n\r\x1b\[0m"

    def script(s):
        s.connect("a"); s.login("a", A, prompt_re=pa)
        s.connect("b"); s.login("b", B, sync_suffix=b"COM> ")
        s.connect("c"); s.login("c", C, colour=True, prompt_re=pc)
        s.line("a", "a prompt follows every line")
        s.line("b", "say hello from command mode")
        s.line("b", "shout and a shout")
        s.line("c", ".tell alice prompts on both ends")
        s.line("b", "mode", sync_suffix=b"")
        s.line("b", "back in speech mode")
        s.line("a", ".prompt", prompt_re=b"")
        s.line("a", "no prompt now")
        s.line("c", ".invis", prompt_re=pc_invis, note="an invisible user's prompt carries a plus sign")
        s.line("c", "still prompted")
        s.line("c", ".vis", prompt_re=pc)
        s.line("b", ".mode", sync_suffix=b"COM> ")
        s.line("b", "invis", sync_suffix=b"COM+> ")
        s.line("b", "look")

    return {}, accounts, script


def afk_bcast():
    """The AFK branch of the main loop, and the two level-scoped fan-outs (nuts333.c:180-203, 4149-4155,
    4772-4788, 6527-6565, 7409-7454)."""
    accounts = [_acc(A), _acc(B), _acc(C, level=2, colour=1), _acc(D, level=3)]

    def script(s):
        for k, n in (("a", A), ("b", B), ("c", C), ("d", D)):
            s.connect(k); s.login(k, n, colour=(k == "c"))    # Carol hears bcast / wizshout with colour on
        s.line("a", ".afk", can_sync=False)
        s.line("b", ".tell alice are you there")
        s.line("b", "< alice pokes")
        s.line("b", ".look")
        s.line("b", "alice still hears the room")
        s.line("a", "typing anything comes back first", can_sync=True)
        s.line("a", ".afk back in five", can_sync=False)
        s.line("b", ".tell alice hello?")
        s.line("a", "", can_sync=True, note="an empty line is enough")
        s.line("a", ".afk lock gone fishing", can_sync=False)
        s.line("a", "wrongpassword", expect=b"Incorrect password.\n\r")
        s.line("a", "test", expect=b"no longer AFK.\n\r", can_sync=True, note="cls, then unlocked")
        s.line("a", ".afk " + "x" * 61)
        s.line("c", ".wizshout for the wizzes")
        s.line("c", ".wizshout ARCH too high for me")
        s.line("d", ".wizshout wiz to WIZ and above")
        s.line("d", ".wizshout arch to ARCH only")
        s.line("d", ".wizshout user levels below WIZ are refused")
        s.line("d", ".wizshout")
        s.line("b", ".wizshout not for mortals")
        s.line("b", ".ignall")
        s.line("c", ".bcast everybody hears this, even those ignoring")
        s.line("d", ".invis")
        s.line("d", ".bcast anonymous")
        s.line("a", ".bcast")
        s.line("a", ".cls")

    return {}, accounts, script


def clones():
    """The clone branch of the fan-out: a listener object in another room that relays to its owner
    (nuts333.c:1416-1426, 4085-4090, 7100-7357, 2870-2882)."""
    accounts = [_acc(A), _acc(B), _acc(D, level=3)]

    def script(s):
        for k, n in (("a", A), ("b", B), ("d", D)):
            s.connect(k); s.login(k, n)
        s.line("a", ".go hallway")
        s.line("d", ".clone hallway")
        s.line("a", ".look", note="the clone is listed like a user")
        s.line("a", "said next to the clone")
        s.line("a", ";emotes next to the clone")
        s.line("a", ".shout shouts are not relayed: the owner hears them anyway")
        s.line("b", "said in the drive, where the owner is")
        s.line("d", ".clone hallway", note="one per room")
        s.line("d", ".clone")
        s.line("d", ".clone corridor", note="max_clones 2")
        s.line("d", ".myclones")
        s.line("d", ".chear hallway swears")
        s.line("a", "a clean line is dropped")
        s.line("a", "a shit line is relayed")
        s.line("d", ".chear hallway nothing")
        s.line("a", "nothing gets through")
        s.line("d", ".chear hallway everything")
        s.line("d", ".chear hallway all")
        s.line("d", ".ignall")
        s.line("a", "owner ignores everyone: no relay")
        s.line("d", ".ignall")
        s.line("d", ".csay hallway the clone speaks")
        s.line("d", ".csay hallway does it ask?")
        s.line("d", ".csay lounge no clone there")
        s.line("a", ".review")
        s.line("a", ".tell dave tells go to the owner, not the clone")
        s.line("d", ".switch hallway")
        s.line("a", "now dave himself is here")
        s.line("b", "and the clone listens in the drive")
        s.line("d", ".destroy drive")
        s.line("d", ".destroy drive")
        s.line("d", ".destroy hallway alice")
        s.line("d", ".clone corridor")
        s.close("d")
        s.line("a", "the owner left, the clones went with him")

    return {"max_clones": 2}, accounts, script


def charecho():
    """Character-mode clients with server-side echo (nuts333.c:369-399, 6881-6893): typed bytes are echoed raw,
    DEL/BS come back as "\\b \\b", the terminator as a newline; nothing is echoed while a password is typed."""
    accounts = [_acc(A, charmode_echo=1), _acc(B)]
    look_end = b"has been set yet.\n\r"
    pw_prompt = b"Give me a password: \xff\xfb\x01"

    def script(s):
        s.connect("b"); s.login("b", B)
        s.connect("a")
        s.raw_dialog("a", [b"ali", b"ce\n"], pw_prompt, note="name typed in pieces: the account's echo flag is not loaded yet")
        s.raw_dialog("a", [b"te", b"st\n"], look_end, logged_in=True, note="password typed in pieces: never echoed")
        s.raw("a", [b"he", b"l\x7flo", b" wor\x08\x08\x08all\n"], note="echo, erase, echo, newline")
        s.raw("a", [b"x", b"\x7f\x7f\x7fy\n"], note="erasing past the start echoes only what was erased")
        s.raw("a", [b"whole line at once\n"], note="a complete line is not character mode: no echo")
        s.line("a", ".charecho")
        s.raw("a", [b"qu", b"iet\n"], note="echo off")
        s.line("b", ".charecho")
        s.raw("b", [b"lo", b"ud\n"], note="echo on for bobby")

    return {}, accounts, script


def rooms():
    """go / move_user / look and who hears what while people move (nuts333.c:3942-4004, 4305-4459)."""
    accounts = [_acc(A), _acc(B, level=2), _acc(C, level=3), _acc(D, in_phrase="bounces in", out_phrase="rolls out")]

    def script(s):
        for k, n in (("a", A), ("b", B), ("c", C), ("d", D)):
            s.connect(k); s.login(k, n)
        s.line("a", ".go")
        s.line("a", ".go nowhere")
        s.line("a", ".go drive")
        s.line("a", ".go lounge", note="not adjoined, and a USER cannot teleport")
        s.line("a", ".go ha", note="room names match by prefix")
        s.line("d", ".go hallway", note="custom in/out phrases")
        s.line("a", ".go wizroom", note="fixed-private: below WIZ stays out")
        s.line("b", ".go wizroom", note="a WIZ teleports to a room that is not adjoined")
        s.line("b", ".go lounge")
        s.line("c", ".invis")
        s.line("c", ".go hallway", note="invisible movement")
        s.line("a", ".look", note="an invisible ARCH is hidden from a USER")
        s.line("c", ".look")
        s.line("b", ".go hallway")
        s.line("b", ".look", note="... and from a WIZ")
        s.line("c", ".vis")
        s.line("a", ".look")
        s.line("a", "said in the hallway")
        s.line("c", ".go drive")
        s.line("c", "alone in the drive")

    return {}, accounts, script


def login_paths():
    """accept_connection + the three login stages, error paths included (nuts333.c:263-311, 1451-1606)."""
    accounts = [_acc(A), _acc(B, level=0), _acc(D, level=4)]
    name_prompt = b"Give me a name: "
    iac_on = b"\xff\xfc\x01"
    pw_prompt = b"Give me a password: \xff\xfb\x01"
    look_end = b"has been set yet.\n\r"

    def script(s):
        s.connect("x")
        s.dialog("x", "", name_prompt, note="empty name")
        s.dialog("x", "version", name_prompt)
        s.dialog("x", "ab", name_prompt + iac_on, note="too short: attempt 1")
        s.dialog("x", "abcdefghijklm", name_prompt + iac_on, note="too long: attempt 2")
        s.dialog("x", "r2d2", closes=True, note="not letters: attempt 3 -> dropped")
        s.connect("x")
        s.dialog("x", "quit", closes=True)
        s.connect("x")
        s.dialog("x", "  alice   trailing words ignored", pw_prompt, note="first word only; case kept, first letter raised")
        s.dialog("x", "ab", name_prompt + iac_on, note="password too short")
        s.dialog("x", "alice", pw_prompt)
        s.dialog("x", "wrongpassword", name_prompt + iac_on, note="incorrect login")
        s.dialog("x", "alice", pw_prompt)
        s.dialog("x", "test", look_end, logged_in=True, note="third attempt succeeds")
        # a hung login of the same name is cleared when somebody else types that name
        s.connect("h")
        s.dialog("h", "bobby", pw_prompt)
        s.connect("y")
        s.dialog("y", "bobby", pw_prompt, note="the other half-open Bobby is dropped")
        s.dialog("y", "test", look_end, logged_in=True)
        # new account
        s.connect("n")
        s.dialog("n", "newbie", b"Give me a password: \xff\xfb\x01", note="unknown name: new user")
        s.dialog("n", "secret", b"confirm password: ")
        s.dialog("n", "secreX", name_prompt + iac_on, note="confirmation mismatch")
        s.dialog("n", "newbie", pw_prompt)
        s.dialog("n", "secret", b"confirm password: ")
        s.dialog("n", "secret", look_end, logged_in=True, note="created at level NEW")
        s.line("n", "a new user may speak")
        s.line("n", ".shout but not shout")
        # wizport
        s.connect("w", wizport=True)
        s.dialog("w", "alice", closes=True, note="USER on the wizport")
        s.connect("w", wizport=True)
        s.dialog("w", "nosuchuser", closes=True, note="no new accounts on the wizport")
        s.connect("w", wizport=True, expect=b"** Wizport login **\n\r\n\r\xff\xfc\x01" + name_prompt)
        s.dialog("w", "dave", pw_prompt)
        s.dialog("w", "test", look_end, logged_in=True)
        # banned name
        s.connect("z")
        s.dialog("z", "Mallory", closes=True, note="listed in datafiles/userban")
        s.close("n")

    return {"configs": lambda p: [pv.TalkerConfig(mainport=p[0][0], wizport=p[0][1], linkport=p[0][2], max_users=50)],
            "accounts": [accounts], "boot_order": [0], "script": script,
            "files": {"datafiles/userban": "Mallory\n"}, "collect_files": ["userfiles/Newbie.D", "userfiles/Alice.D"]}


def capacity():
    """max_users applies to the main port only, counting half-open logins (nuts333.c:287-291)."""
    accounts = [_acc(A), _acc(B), _acc(D, level=4)]
    look_end = b"has been set yet.\n\r"
    pw_prompt = b"Give me a password: \xff\xfb\x01"

    def script(s):
        s.connect("a"); s.login("a", A)
        s.connect("h")                                  # half-open login: counts
        s.connect("c", closes=True)                     # third: full
        s.connect("w", wizport=True, expect=b"Give me a name: ")   # the wizport is exempt
        s.dialog("w", "dave", pw_prompt)
        s.dialog("w", "test", look_end, logged_in=True)
        s.line("a", "two users and a half-open login")

    return {"max_users": 2}, accounts, script


def netlink():
    """talker 0 ("talker1") dials talker 1 ("talker2") at boot; Alice travels there and back."""
    acc1 = [_acc(A), _acc(D)]
    acc2 = [_acc(B), _acc(C, colour=1)]

    def configs(p):
        rooms1 = tuple(pv.Room(r.label, r.name, r.links, r.access, "CONNECT talker2" if r.name == "drive" else "",
                               r.description) for r in pv.DEFAULT_ROOMS)
        cfg1 = pv.TalkerConfig(mainport=p[0][0], wizport=p[0][1], linkport=p[0][2], max_users=50, verification="verify1",
                               auto_connect=True, rooms=rooms1, sites=[pv.Site("talker2", "127.0.0.1", p[1][2], "verify2")])
        # the accepting side matches the dialler by its reverse-resolved site string
        # (nuts333.c:322, 2908-2909): list both spellings
        cfg2 = pv.TalkerConfig(mainport=p[1][0], wizport=p[1][1], linkport=p[1][2], max_users=50, verification="verify2",
                               sites=[pv.Site("talker1", "localhost", p[0][2], "verify1"),
                                      pv.Site("talker1", "127.0.0.1", p[0][2], "verify1")])
        return [cfg1, cfg2]

    look_end = b"has been set yet.\n\r"

    def script(s):
        s.connect("a", talker=0); s.login("a", A)
        s.connect("d", talker=0); s.login("d", D)
        s.connect("b", talker=1); s.login("b", B)
        s.connect("c", talker=1); s.login("c", C, colour=True)
        for hop in pv.WALKS["lounge"]:                 # the ACCEPT room of talker2 is its lounge
            s.line("b", f".go {hop}")
            s.line("c", f".go {hop}")
        s.line("a", "said on talker1 before leaving")
        s.line("a", ".go talker2", expect=look_end, note="TRANS -> GRANTED -> ACT look -> MSG frames")
        s.line("a", "hello from afar")
        s.line("a", "a question from afar?")
        s.line("a", ".shout shouting on talker2")
        s.line("b", ".shout bobby shouts back")
        s.line("b", ".tell alice a tell across the link")
        s.line("a", ".tell bobby and an answer?")
        s.line("a", ";waves from afar")
        s.line("c", "~FRcoloured~RS say reaches alice stripped at home")
        s.line("d", ".tell alice are you there")
        s.line("d", ".shout heard on talker1 only")
        s.line("a", ".look")
        s.line("a", ".colour", colour=True, note="home execution")
        s.line("c", "~FRcoloured~RS say reaches alice in colour now")
        s.line("a", ".colour", colour=False)
        s.line("a", ".go talker1", expect=look_end, note="REMVD: back home")
        s.line("a", "home again")
        s.line("b", ".shout alice no longer hears talker2")

    return {"configs": configs, "accounts": [acc1, acc2], "boot_order": [1, 0],
            "wait_syslog": [(0, "Connection to talker2 verified")], "script": script}


def netlink_wire_accept():
    """A scripted peer dials the talker and speaks the netlink protocol to it, verb by verb."""
    accounts = [_acc(B), _acc(C)]

    def configs(p, peer_ports):
        return [pv.TalkerConfig(mainport=p[0][0], wizport=p[0][1], linkport=p[0][2], max_users=50, verification="verify0",
                                sites=[pv.Site("peer1", "localhost", 1, "verify1"), pv.Site("peer1", "127.0.0.1", 1, "verify1")])]

    def script(s):
        s.connect("b"); s.login("b", B)
        for hop in pv.WALKS["lounge"]:
            s.line("b", f".go {hop}")
        s.peers["p"].dial(s.link_ports[0])
        s.peer_step("p", b"", b"GRANTED CONNECT\n", note="accept_server_connection: version greeting, then grant")
        s.peer_step("p", b"VERIFICATION verify1 3.3.3\n", b"VERIFY OK ALL\n")
        s.peer_step("p", b"TRANS Alice NUKyNCCLvgLH. 1 is alice\n", b"GRANTED Alice\n", note="no local account: description and level come from the frame")
        s.peer_step("p", b"ACT Alice look\n", b"PRM Alice\n", note="one MSG..EMSG frame per write_user call, then PRM")
        s.peer_step("p", b"ACT Alice say hello there\n", b"PRM Alice\n")
        s.peer_step("p", b"ACT Alice .shout loud and clear\n", b"PRM Alice\n")
        s.peer_step("p", b"ACT Alice .tell bobby psst?\n", b"PRM Alice\n")
        s.line("b", ".tell alice back at you"); s.peer_expect("p", b"EMSG\n")
        s.line("b", "a room say reaches the remote user"); s.peer_expect("p", b"EMSG\n")
        s.line("b", "~FRmarkup~RS crosses the link unexpanded"); s.peer_expect("p", b"EMSG\n")
        s.peer_step("p", b"ACT Alice NL\nACT Alice .version\n", b"PRM Alice\n", note="two frames in one segment; NL produces nothing")
        s.peer_step("p", b"ACT Alice say par", None, note="a line split over two segments is buffered")
        s.peer_step("p", b"tial line\n", b"PRM Alice\n")
        s.peer_step("p", b"KA\nFOO bar\n", b"ERROR\n", note="keepalive is silent; an unknown verb is answered with ERROR")
        s.peer_step("p", b"TRANS Alice x 1 again\n", b"DENIED Alice 5\n", note="already here")
        s.peer_step("p", b"ACT Nobody look\n", b"DENIED Nobody 8\n")
        s.peer_step("p", b"TRANS Carol wronghash 1 has an account here\n", b"DENIED Carol 7\n", note="local account, password mismatch")
        s.peer_step("p", b"TRANS Dave somehash 4 is a god elsewhere\n", b"GRANTED Dave\n", note="level capped at rem_user_maxlevel")
        s.peer_step("p", b"ACT Dave .invis\n", b"PRM Dave\n", note="capped to WIZ: .invis needs ARCH")
        s.peer_step("p", b"ACT Dave .passwd\n", b"PRM Dave\n", note="barred for remote users")
        s.peer_step("p", b"MSG Bobby\nrelayed line one\nrelayed line two\nEMSG\nACT Alice .version\n", b"PRM Alice\n",
                    note="MSG body is relayed line by line to the named user")
        s.peer_step("p", b"REL Dave\nACT Alice .version\n", b"PRM Alice\n")
        s.peer_step("p", b"DISCONNECT\n", None, closes=True, note="remote users vanish, the link is closed")
        s.line("b", "after the link went down")

    return {"configs": configs, "accounts": [accounts], "boot_order": [0], "peers": ["p"], "script": script}


def netlink_wire_dial():
    """The talker dials a scripted peer at boot (auto_connect) and a local user travels out and back."""
    accounts = [_acc(A), _acc(D)]
    look_end = b"has been set yet.\n\r"

    def configs(p, peer_ports):
        rooms1 = tuple(pv.Room(r.label, r.name, r.links, r.access, "CONNECT peer2" if r.name == "drive" else "",
                               r.description) for r in pv.DEFAULT_ROOMS)
        return [pv.TalkerConfig(mainport=p[0][0], wizport=p[0][1], linkport=p[0][2], max_users=50, verification="verify1",
                                auto_connect=True, rooms=rooms1, sites=[pv.Site("peer2", "127.0.0.1", peer_ports["p"], "verify2")])]

    def script(s):
        p = s.peers["p"]
        p.accept()
        s.connect("a"); s.login("a", A)
        s.connect("d"); s.login("d", D)
        s.peer_step("p", b"NUTS 3.3.3\nGRANTED CONNECT\n", b"\n", note="the dialler answers the grant with its verification")
        s.peer_step("p", b"VERIFY OK ALL\n", None, note="link up: announced to everyone")
        s.line("a", ".look")
        s.send_only("a", ".go peer2", {"p": b"\n"}, note="TRANS <name> <hash> <level> <desc>")
        s.set_flags("a", can_sync=False)
        s.peer_step("p", b"GRANTED Alice\n", b"ACT Alice look\n", client_expect={"a": b"cyberspace...\n\r"})
        s.peer_step("p", b"MSG Alice\n\n~FTRoom: ~FGelsewhere\n\nEMSG\nMSG Alice\nno newline at the end of this frame\nEMSG\nPRM Alice\n", None,
                    client_expect={"a": b"end of this frame\n\r"})
        s.send_only("a", "hello remote", {"p": b"\n"})
        s.send_only("a", "is this relayed?", {"p": b"\n"})
        s.send_only("a", ".shout x y", {"p": b"\n"})
        s.send_only("a", ";emotes", {"p": b"\n"})
        s.send_only("a", "> bobby hi", {"p": b"\n"})
        s.send_only("a", "", {"p": b"\n"}, note="an empty line travels as NL")
        s.send_only("a", ".bogus", {"p": b"\n"}, note="unknown commands are relayed too: the remote decides")
        s.line("d", ".tell alice are you there")
        s.peer_step("p", b"REMVD Alice\n", None, client_expect={"a": look_end}, note="sent home")
        s.set_flags("a", can_sync=True)
        s.line("a", "back home")
        s.send_only("a", ".go peer2 secret", {"p": b"\n"}, note="explicit remote password is crypt()ed into the frame")
        s.peer_step("p", b"DENIED Alice 7\n", None, client_expect={"a": b"password>'.\n\r"})
        s.send_only("a", ".go peer2", {"p": b"\n"})
        s.line("a", ".go peer2", note="still waiting for the grant")
        s.line("a", ".go hallway", expect=look_end, note="gives up: REL goes out")
        s.peer_expect("p", b"\n")
        s.line("a", ".go drive")
        s.peer_step("p", b"DISCONNECT\n", None, closes=True)
        s.line("a", ".go peer2")

    return {"configs": configs, "accounts": [accounts], "boot_order": [0], "peers": ["p"],
            "wait_syslog": [(0, "Connected to peer2")], "script": script}


#: banners over 1 KB whose newlines, colour commands and fill level land on the staging buffer's three flush rules
#: (nuts333.c:2253-2255, 2272-2274, 2292-2294); shared with tests/test_harness.py, which compares write(2) counts
LONG_MOTD1 = ("a" * 995 + "\n"                      # newline with 995 staged (> 994): flush first
              + "b" * 993 + "~FRx\n"                 # '~' with 995 staged: flush first; colour off: the code vanishes
              + "c" * 997 + "\n"                     # fills the buffer to exactly 1000: flush
              + "/~FR escaped, ~ZZ unknown, the end of motd1\n\n")
LONG_MOTD2 = ("a" * 990 + "~OLbbb\n"                 # colour on: 990 + ESC[1m + bbb = 997 staged at the newline
              + "c" * 988 + "~FGd" + "e" * 30 + "~RS\n")   # 6 + 988 + ESC[32m = 999, 'd' makes 1000


def long_motd():
    """Pre- and post-login banners longer than the 1000-byte staging buffer of more() (nuts333.c:2205-2296), read
    by a colour-on user: pins the bytes; the write(2) boundaries are compared in tests/test_harness.py."""
    accounts = [_acc(A, colour=1), _acc(B)]

    def script(s):
        s.connect("a"); s.login("a", A, colour=True)
        s.connect("b"); s.login("b", B)
        s.line("a", "after the long banners")

    return {"configs": lambda p: [pv.TalkerConfig(mainport=p[0][0], wizport=p[0][1], linkport=p[0][2], max_users=50)],
            "accounts": [accounts], "boot_order": [0], "script": script,
            "files": {"motd1": LONG_MOTD1, "motd2": LONG_MOTD2}}


def netlink_wire_legacy():
    """Peers that announce an older protocol version get the older answers (nuts333.c:1299-1300, 3093-3097, 3129-3139,
    3141-3146): a banned or locked-out traveller is DENIED 6 instead of 9 / 8, a pre-3.3.1 TRANS carries no level word
    and the user gets rem_user_deflevel, a pre-3.2 peer is sent MSG bodies with the colour commands stripped.  Three
    links in a row on the same talker: 3.3.2, 3.1.0, then 3.3.3 for the contrast."""
    accounts = [_acc(B)]

    def configs(p, peer_ports):
        return [pv.TalkerConfig(mainport=p[0][0], wizport=p[0][1], linkport=p[0][2], max_users=50, verification="verify0",
                                minlogin_level="USER", rem_user_deflevel="WIZ",
                                sites=[pv.Site("peer1", "localhost", 1, "verify1"), pv.Site("peer1", "127.0.0.1", 1, "verify1")])]

    def link_up(s, key, version):
        s.peers[key].dial(s.link_ports[0])
        s.peer_step(key, b"", b"GRANTED CONNECT\n")
        s.peer_step(key, b"VERIFICATION verify1 " + version + b"\n", b"VERIFY OK ALL\n")

    def script(s):
        s.connect("b"); s.login("b", B)
        for hop in pv.WALKS["lounge"]:
            s.line("b", f".go {hop}")
        link_up(s, "p", b"3.3.2")
        s.peer_step("p", b"TRANS Mallory somehash 1 is banned here\n", b"DENIED Mallory 6\n", note="banned: the old code for an old peer")
        s.peer_step("p", b"TRANS Newbie somehash 0 is below minlogin_level\n", b"DENIED Newbie 6\n", note="locked out: the old code too")
        s.peer_step("p", b"TRANS Alice somehash 4 has a level word\n", b"GRANTED Alice\n", note="3.3.1+: level taken from the frame, capped")
        s.peer_step("p", b"ACT Alice .wizshout capped at WIZ\n", b"PRM Alice\n")
        s.line("b", "~FRcolour commands~RS cross a 3.3.2 link as they are"); s.peer_expect("p", b"EMSG\n")
        s.peer_step("p", b"DISCONNECT\n", None, closes=True)
        link_up(s, "q", b"3.1.0")
        s.peer_step("q", b"TRANS Carol somehash the description starts at the third word\n", b"GRANTED Carol\n",
                    note="pre-3.3.1: no level word; the user gets rem_user_deflevel")
        s.peer_step("q", b"ACT Carol .wizshout deflevel is WIZ here\n", b"PRM Carol\n")
        s.line("b", ".look", note="the description as the talker took it")
        s.line("b", "~FRcolour commands~RS are stripped for a pre-3.2 peer, /~FR too"); s.peer_expect("q", b"EMSG\n")
        s.peer_step("q", b"TRANS Mallory somehash banned\n", b"DENIED Mallory 6\n")
        s.peer_step("q", b"DISCONNECT\n", None, closes=True)
        link_up(s, "r", b"3.3.3")
        s.peer_step("r", b"TRANS Mallory somehash 1 is banned here\n", b"DENIED Mallory 9\n", note="3.3.3 peers get the new codes")
        s.peer_step("r", b"TRANS Newbie somehash 0 is below minlogin_level\n", b"DENIED Newbie 8\n")
        s.peer_step("r", b"DISCONNECT\n", None, closes=True)
        s.line("b", "all three links are gone")

    return {"configs": configs, "accounts": [accounts], "boot_order": [0], "peers": ["p", "q", "r"], "script": script,
            "files": {"datafiles/userban": "Mallory\n"}}


def board_mail_files():
    """REFERENCE ONLY (the restatement answers board and mail commands with a notice): the two
    remaining on-disk formats of SURVEY.md 8(f)4 -- board ``.B`` (``PT:`` header, 80-column wrap, nuts333.c:5008-5040)
    and mail ``.M`` (time stamp on the first line, older mail kept below it, nuts333.c:2462-2503)."""
    accounts = [_acc(A), _acc(B), _acc(D, level=3)]

    def script(s):
        for k, n in (("a", A), ("b", B), ("d", D)):
            s.connect(k); s.login(k, n)
        s.line("a", ".write first message on the board")
        s.line("b", ".write a second one ~FRwith a colour command~RS kept as typed")
        s.line("a", ".write " + "w" * 170, note="the body is wrapped after 80 characters")
        s.line("d", ".invis")
        s.line("d", ".write from a presence", note="an invisible writer is not named")
        s.line("a", ".read")
        s.line("a", ".smail bobby hello by mail")
        s.line("d", ".smail bobby a second mail: the new stamp goes on top, the old mail stays")
        s.line("b", ".rmail")
        s.line("a", ".smail nobody no such user")

    return {"configs": lambda p: [pv.TalkerConfig(mainport=p[0][0], wizport=p[0][1], linkport=p[0][2], max_users=50)],
            "accounts": [accounts], "boot_order": [0], "script": script,
            "collect_files": ["datafiles/drive.B", "userfiles/Bobby.M"]}


#: scenarios only the reference can run; fixtures under tests/golden/reference_only/
REFERENCE_ONLY = {"board_mail_files": board_mail_files}

SCENARIOS = {
    "speech_colour_off": speech_colour_off,
    "speech_colour_mixed": speech_colour_mixed,
    "markup": markup,
    "filters": filters,
    "errors": errors,
    "swearing": swearing,
    "framing": framing,
    "review": review,
    "prompts": prompts,
    "afk_bcast": afk_bcast,
    "charecho": charecho,
    "clones": clones,
    "rooms": rooms,
    "login_paths": login_paths,
    "capacity": capacity,
    "netlink": netlink,
    "netlink_wire_accept": netlink_wire_accept,
    "netlink_wire_dial": netlink_wire_dial,
    "netlink_wire_legacy": netlink_wire_legacy,
    "long_motd": long_motd,
}
