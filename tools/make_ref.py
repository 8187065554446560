#!/usr/bin/env python
"""make_ref.py — build oracle/_ref: the REFERENCE'S OWN decision code, compiled (test infrastructure).

The reference (curioloop/rafting) is Java and there is no JDK here, so it cannot run.  Its decision classes are,
however, plain imperative Java over long/boolean fields.  This script translates them MECHANICALLY — token by token,
no hand-edited output — into C++ that compiles against a small stand-in for the JDK (oracle/ref_shim/jrt.hpp) and for
the reference's I/O plugins (oracle/ref_shim/env.hpp).  The result, oracle/_ref/libref.so, executes the reference's
own control flow (every `if`, every assertion message, every call order); tests/test_ref_parity.py differentially
tests the hand-written oracle (oracle/raft_oracle.c) against it.

What is translated (MANIFEST below; sha256 of every source range is pinned — a changed reference fails loudly):
  whole classes   RaftResponse, RaftParticipant, RaftLog (+Entry, EntryKey), RocksEntry, Membership, Leadership (+State),
                  RaftMember, TimerTicket, Follower, Candidate, Leader
  members         RocksLog: every log operation (newEntry .. lastEntry, longToBytes/bytesToLong) over a fake RocksDB
  methods         RaftRoutine: keepAlive, electionTimeout, resetTimer, trySwitch, switchTo, convertTo
                  RaftContext: accessors, resetTimer, switchTo, trySwitchTo, acceptCommand, commitLog
  lambda bodies   the three response callbacks of Leader.replicateLog / Candidate.startElection / Follower.prepareElection
                  are ALSO lifted into named methods (captured variables become parameters) so a test can deliver a
                  response with arbitrary closure state; the original lambdas stay in place and are used as well.

Translation rules (all mechanical; see the functions below): `.` -> `->` on references (null-checked), static access
-> `::`, reference types -> Ref<T>, arrays -> JArr<T>, `new T(..)` -> jnew<T>(..), lambdas -> C++ lambdas capturing by
copy, `try/finally` -> scope guard, `instanceof` / casts -> helpers, `X.class` -> X_class, fields whose name collides
with a method (legal in Java, not in C++) get a trailing underscore, declarations and definitions are split so
classes may refer to each other freely.  Two substitutions replace Java-only constructs (listed in SUBSTITUTIONS).

Outputs go ONLY to oracle/_ref/gen/ (git-ignored: reference-derived text never enters the repository).
usage: python tools/make_ref.py [--ref /root/reference] [--out oracle/_ref/gen] [--print-sha]
"""
import argparse
import hashlib
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "src/main/java/io/lubricant/consensus/raft"

# ------------------------------------------------------------------------------------------------------------------
# manifest

MANIFEST = [
    dict(file="RaftResponse.java", mode="class", exclude=["toString"]),
    dict(file="RaftParticipant.java", mode="class"),
    dict(file="command/RaftLog.java", mode="class"),
    dict(file="command/storage/RocksEntry.java", mode="class", exclude=["toString"]),
    dict(file="context/member/Membership.java", mode="class"),
    dict(file="context/member/Leadership.java", mode="class"),
    dict(file="context/member/RaftMember.java", mode="class"),
    dict(file="context/member/TimerTicket.java", mode="class"),
    dict(file="context/member/Follower.java", mode="class",
         lift=[dict(name="onPreVoteResponse", lines=(258, 270),
                    params="ID id, long nextTerm, AtomicInteger votes, int majority, AsyncHead head, "
                           "RaftResponse result, Throwable error, boolean canceled")]),
    dict(file="context/member/Candidate.java", mode="class",
         lift=[dict(name="onRequestVoteResponse", lines=(121, 134),
                    params="ID id, AtomicInteger votes, int majority, AsyncHead head, "
                           "RaftResponse result, Throwable error, boolean canceled")]),
    dict(file="context/member/Leader.java", mode="class",
         lift=[dict(name="onInstallSnapshotResponse", lines=(174, 188),
                    params="ID id, State state, Entry epoch, AsyncHead head, "
                           "RaftResponse result, Throwable error, boolean canceled"),
               dict(name="onAppendEntriesResponse", lines=(218, 237),
                    params="ID id, State state, Entry epoch, long lastIndex, long nextIndex, AsyncHead head, "
                           "RaftResponse result, Throwable error, boolean canceled")]),
    dict(file="command/storage/RocksLog.java", mode="class", keep=[(45, 53), (81, 253), (259, 280)],
         extra_members="""
    // (skeleton, not reference text) the constructor opens a real RocksDB in the reference (storage/RocksLog.java:55-79)
    static inline JArr<jbyte> INDEX = jbytes("index"), TERM = jbytes("term");
    RocksLog(Ref<RocksDB> db_, Ref<RocksSerializer> ser_);
""",
         extra_defs="""
inline RocksLog::RocksLog(Ref<RocksDB> db_, Ref<RocksSerializer> ser_)
{
    db = db_; serializer = ser_; epoch_ = jnew<ColumnFamilyHandle>();
    lastEntry_ = lastEntry();                                     // storage/RocksLog.java:69
    epochEntry = jnew<EntryKey>((jlong)0, (jlong)0);              // :74-75
}
"""),
    # N2: the wire frame grammar SOH|TYPE|{SEQ}|STX|HEADLEN|HEAD|BODYLEN|{BODY}|ETX[|EOT] — constants, EventFrame, the streaming
    # FrameDecoder state machine and FrameEncoder.encode, over a ByteBuf stand-in (oracle/ref_shim/wire_env.hpp)
    dict(file="transport/EventCodec.java", mode="class", keep=[(25, 51), (169, 196), (219, 335)], exclude=["allocateBuffer"],
         hoist_statics=True, unit="wire"),
    dict(file="context/RaftRoutine.java", mode="methods", cls="RaftRoutine",
         ranges=[(53, 62), (65, 77), (86, 130), (140, 152), (159, 181), (183, 216)]),
    dict(file="context/RaftContext.java", mode="methods", cls="RaftContext",
         ranges=[(165, 166), (168, 170), (172, 173), (175, 180), (185, 187), (195, 197), (205, 215), (223, 237),
                 (244, 255), (260, 262)],
         collisions=["cluster", "envConfig", "replicatedLog", "stateMachine", "stableStorage", "snapArchive",
                     "eventLoop", "stillRunning"]),
]

# Java-only constructs replaced by a shim call.  (file, first line, last line, replacement C++ tokens)
SUBSTITUTIONS = [
    # an anonymous ConcurrentHashMap subclass whose values() returns a fixed-order list (member/Leader.java:44-49)
    ("context/member/Leader.java", 44, 49, "followerStatus = jrt_fixed_values_map<ID, State>(map, states);"),
    # reflection: getConstructor(...).newInstance(...) over the role class (context/RaftRoutine.java:203-205)
    ("context/RaftRoutine.java", 203, 205,
     "participant = member->role()->newInstance(context, member->term(), member->ballot(), member);"),
]

PRIMS = {"long": "jlong", "int": "jint", "boolean": "jboolean", "byte": "jbyte", "short": "jshort", "double": "jdouble",
         "void": "void", "char": "jchar", "float": "jfloat"}
DROP_MODIFIERS = {"public", "private", "protected", "final", "volatile", "synchronized", "abstract", "transient", "default"}
DROP_BASES = {"Serializable", "Closeable", "AutoCloseable", "Comparable", "MessageToByteEncoder", "ByteToMessageDecoder"}
GENERIC_TYPES = {"Map", "HashMap", "ConcurrentHashMap", "Set", "List", "ArrayList", "Collection", "Async", "AtomicReference",
                 "AtomicLongFieldUpdater", "AtomicIntegerFieldUpdater", "Class", "ScheduledFuture", "Function", "Future",
                 "Promise", "PendingTask", "CompletableFuture"}
RAW_GENERICS = {"Class", "ScheduledFuture", "Promise", "PendingTask"}    # generic arguments are dropped
# identifiers that name a type when they start a declaration (translated classes are added at run time)
KNOWN_TYPES = set(GENERIC_TYPES) | {
    "ID", "Command", "String", "Object", "Throwable", "Exception", "AtomicInteger", "AtomicLong", "AsyncHead", "RaftService",
    "RaftCluster", "RaftConfig", "RaftMachine", "StableLock", "SnapshotArchive", "ContextEventLoop", "RocksDB",
    "RocksIterator", "ColumnFamilyHandle", "RocksSerializer", "RocksStateLoader", "Path", "Boolean", "Logger",
    "ScheduledExecutorService", "ExecutorService", "RaftContext", "RaftRoutine", "EntryKey", "Entry", "State", "Snapshot",
    "NotLeaderException", "ObsoleteContextException", "ByteBuf", "ChannelHandlerContext", "CharSequence", "SerializeException"}
# identifiers followed by `.` that denote static access (`::`)
STATIC_SCOPES = {"Math", "Long", "Integer", "Arrays", "Objects", "System", "Async", "TimeUnit", "Boolean", "CompletableFuture",
                 "RocksDB", "Collections", "String", "StandardCharsets", "Serialization"}
QUALIFIER_ONLY = {"RaftLog", "RaftCluster", "RaftStub", "Leadership_"}   # `RaftLog.Entry` -> `Entry` (nested types are flattened)
EXC_TYPES = {"Exception", "Throwable", "IOException", "InterruptedException", "RuntimeException", "AssertionError"}


def die(msg):
    sys.exit("make_ref: " + msg)


# ------------------------------------------------------------------------------------------------------------------
# tokens

class Tok:
    __slots__ = ("k", "t", "ws", "line")

    def __init__(self, k, t, ws, line):
        self.k, self.t, self.ws, self.line = k, t, ws, line

    def __repr__(self):
        return "%s:%r" % (self.k, self.t)


TOKEN_RE = re.compile(r"""
    (?P<ws>\s+)
  | (?P<lc>//[^\n]*)
  | (?P<bc>/\*.*?\*/)
  | (?P<str>"(?:\\.|[^"\\])*")
  | (?P<chr>'(?:\\.|[^'\\])*')
  | (?P<num>\d[\dA-Fa-fxXlL_]*(?:\.\d+)?)
  | (?P<id>[A-Za-z_$][A-Za-z_$0-9]*)
  | (?P<op>>>>=|>>>|<<=|>>=|->|::|\+\+|--|&&|\|\||==|!=|<=|>=|\+=|-=|\*=|/=|&=|\|=|\^=|%=|<<|[{}()\[\];,.@=<>!~?:+\-*/&|^%])
""", re.X | re.S)


def tokenize(text, first_line=1):
    out, ws, line, pos = [], "", first_line, 0
    while pos < len(text):
        m = TOKEN_RE.match(text, pos)
        if not m:
            die("cannot tokenize at line %d: %r" % (line, text[pos:pos + 30]))
        kind = m.lastgroup
        s = m.group()
        if kind in ("ws", "lc", "bc"):
            ws += "\n" * s.count("\n") if kind != "ws" else s     # comments vanish, their newlines stay
            if kind == "lc":
                pass
        else:
            out.append(Tok(kind, s, ws, line))
            ws = ""
        line += s.count("\n")
        pos = m.end()
    return out


def render(toks):
    return "".join(t.ws + t.t for t in toks)


def T(k, t, ws=" "):
    return Tok(k, t, ws, 0)


def match_close(toks, i, open_t, close_t):
    """index of the token closing the bracket opened at i"""
    depth = 0
    for j in range(i, len(toks)):
        if toks[j].t == open_t:
            depth += 1
        elif toks[j].t == close_t:
            depth -= 1
            if depth == 0:
                return j
    die("unbalanced %s at line %d" % (open_t, toks[i].line))


# ------------------------------------------------------------------------------------------------------------------
# types

def parse_type(toks, i, known):
    """Parse a Java type starting at toks[i]. Returns (cpp_text, bare_text, next_index) or None."""
    if i >= len(toks) or toks[i].k != "id":
        return None
    name = toks[i].t
    j = i + 1
    # qualified nested names: RaftLog.Entry -> Entry
    while j + 1 < len(toks) and toks[j].t == "." and toks[j + 1].k == "id" and \
            (name in QUALIFIER_ONLY or (toks[j + 1].t in known and toks[j + 1].t[0].isupper() and toks[j + 1].t not in STATIC_SCOPES)):
        name = toks[j + 1].t
        j += 2
    if name in PRIMS:
        bare = cpp = PRIMS[name]
        is_prim = True
    elif name in known:
        is_prim = False
        bare = name
        if j < len(toks) and toks[j].t == "<" and name in GENERIC_TYPES:
            args, j2 = parse_generic_args(toks, j, known)
            if args is None:
                return None
            j = j2
            if name not in RAW_GENERICS and args != []:
                bare = "%s<%s>" % (name, ", ".join(args))
            elif args == []:
                bare = name + "<>"          # diamond: resolved by the caller
        cpp = "JString" if bare == "String" else "Ref<%s>" % bare
    else:
        return None
    while j + 1 < len(toks) and toks[j].t == "[" and toks[j + 1].t == "]":
        cpp = bare = "JArr<%s>" % cpp
        j += 2
    return cpp, bare, j


def parse_generic_args(toks, i, known):
    """toks[i] == '<'. Returns (list of bare element types, index after '>') or (None, i)."""
    j = i + 1
    args = []
    if toks[j].t == ">":
        return [], j + 1
    while True:
        if toks[j].t == "?":                      # wildcard: `? extends X` / `?`
            j += 1
            if toks[j].t in ("extends", "super"):
                r = parse_type(toks, j + 1, known)
                if r is None:
                    return None, i
                j = r[2]
            args.append("?")
        else:
            r = parse_type(toks, j, known)
            if r is None:
                return None, i
            cpp, bare, j = r
            args.append(bare if not bare.startswith("JArr<") and bare not in PRIMS.values() else cpp)
        if toks[j].t == ",":
            j += 1
            continue
        if toks[j].t == ">":
            return args, j + 1
        return None, i


# ------------------------------------------------------------------------------------------------------------------
# class structure

class Member:
    pass


class Field(Member):
    def __init__(self, toks, static, line):
        self.toks, self.static, self.line = toks, static, line


class Method(Member):
    def __init__(self, name, ret, params, body, static, is_ctor, line, has_body):
        self.name, self.ret, self.params, self.body = name, ret, params, body
        self.static, self.is_ctor, self.line, self.has_body = static, is_ctor, line, has_body


class ClassNode(Member):
    def __init__(self, kind, name, extends, implements, members, line):
        self.kind, self.name, self.extends, self.implements, self.members, self.line = kind, name, extends, implements, members, line


def skip_annotation(toks, i):
    while i < len(toks) and toks[i].t == "@":
        i += 2
        if i < len(toks) and toks[i].t == "(":
            i = match_close(toks, i, "(", ")") + 1
    return i


def parse_class(toks, i):
    """toks[i] is 'class' or 'interface'. Returns (ClassNode, index after closing brace)."""
    kind = toks[i].t
    name = toks[i + 1].t
    line = toks[i].line
    j = i + 2
    if toks[j].t == "<":
        j = match_close(toks, j, "<", ">") + 1
    extends, implements = [], []
    while toks[j].t != "{":
        if toks[j].t == "extends":
            tgt = extends if kind == "class" else implements
        elif toks[j].t == "implements":
            tgt = implements
        elif toks[j].k == "id":
            nm = toks[j].t
            while toks[j + 1].t == ".":
                nm = toks[j + 2].t
                j += 2
            if toks[j + 1].t == "<":
                j = match_close(toks, j + 1, "<", ">")
            tgt.append(nm)
        j += 1
    end = match_close(toks, j, "{", "}")
    members = parse_members(toks, j + 1, end, name)
    return ClassNode(kind, name, extends, implements, members, line), end + 1


def parse_members(toks, i, end, cls_name):
    members = []
    while i < end:
        i = skip_annotation(toks, i)
        if i >= end:
            break
        start = i
        static = False
        while toks[i].t in DROP_MODIFIERS or toks[i].t == "static" or toks[i].t == "@":
            if toks[i].t == "static":
                static = True
            if toks[i].t == "@":
                i = skip_annotation(toks, i)
                continue
            i += 1
        if toks[i].t in ("class", "interface"):
            node, i = parse_class(toks, i)
            members.append(node)
            continue
        if toks[i].t == "<":                                   # generic method: not translated
            i = match_close(toks, i, "<", ">") + 1
        # field or method: scan to '=' / ';' / '(' at depth 0
        j = i
        while toks[j].t not in ("=", ";", "("):
            j += 1
        if toks[j].t == "(":
            name_i = j - 1
            close = match_close(toks, j, "(", ")")
            k = close + 1
            if toks[k].t == "throws":
                while toks[k].t not in ("{", ";"):
                    k += 1
            ret = toks[i:name_i]
            params = toks[j + 1:close]
            if toks[k].t == "{":
                bend = match_close(toks, k, "{", "}")
                body = toks[k:bend + 1]
                nxt = bend + 1
                has_body = True
            else:
                body, nxt, has_body = [], k + 1, False
            nm = toks[name_i].t
            members.append(Method(nm, ret, params, body, static, nm == cls_name and not ret, toks[start].line, has_body))
            i = nxt
        else:
            k = j
            depth = 0
            while not (toks[k].t == ";" and depth == 0):
                if toks[k].t in "({[":
                    depth += 1
                elif toks[k].t in ")}]":
                    depth -= 1
                k += 1
            members.append(Field(toks[i:k + 1], static, toks[start].line))
            i = k + 1
    return members


# ------------------------------------------------------------------------------------------------------------------
# body rewriting

class Ctx:
    """what the body rewriter needs to know about its surroundings"""

    def __init__(self, cls, collide_here, collide_any, known, static_scopes, where="?"):
        self.cls, self.collide_here, self.collide_any = cls, collide_here, collide_any
        self.known, self.static_scopes, self.where = known, static_scopes, where


def split_params(params):
    out, cur, depth = [], [], 0
    for t in params:
        if t.t in "<([":
            depth += 1
        elif t.t in ">)]":
            depth -= 1
        if t.t == "," and depth == 0:
            out.append(cur)
            cur = []
        else:
            cur.append(t)
    if cur:
        out.append(cur)
    return out


def cpp_params(params, known):
    """-> (cpp text, [names])"""
    parts, names = [], []
    for p in split_params(params):
        p = [t for t in p if t.t not in DROP_MODIFIERS]
        r = parse_type(p, 0, known)
        if r is None or r[2] != len(p) - 1:
            die("cannot parse parameter %r (line %d)" % (render(p), p[0].line))
        parts.append("%s %s" % (r[0], p[-1].t))
        names.append(p[-1].t)
    return ", ".join(parts), names


def rewrite_finally(toks):
    """try {A} [catch..] finally {B}  ->  { auto _fin = jfinally([&]{B}); try {A} [catch..] }  (plain {A} without catches)"""
    n = [0]

    def one(toks):
        i = 0
        while i < len(toks):
            if toks[i].t == "try" and toks[i + 1].t == "{":
                a_end = match_close(toks, i + 1, "{", "}")
                k = a_end + 1
                catches_from = k
                while k < len(toks) and toks[k].t == "catch":
                    k = match_close(toks, k + 1, "(", ")") + 1
                    k = match_close(toks, k, "{", "}") + 1
                catches = toks[catches_from:k]
                if k < len(toks) and toks[k].t == "finally":
                    b_end = match_close(toks, k + 1, "{", "}")
                    n[0] += 1
                    a = one(toks[i + 1:a_end + 1])
                    b = one(toks[k + 1:b_end + 1])
                    guard = [T("op", "{", toks[i].ws), T("id", "auto"), T("id", "_fin%d" % n[0]), T("op", "="),
                             T("id", "jfinally"), T("op", "(", ""), T("op", "[", ""), T("op", "&", ""), T("op", "]", "")] + b + \
                            [T("op", ")", ""), T("op", ";", "")]
                    if catches:
                        inner = [T("id", "try")] + a + one(catches)
                    else:
                        inner = a
                    return toks[:i] + guard + inner + [T("op", "}")] + one(toks[b_end + 1:])
            i += 1
        return toks
    return one(toks)


def rewrite_body(toks, ctx, locals_):
    """Token-level Java -> C++ for a statement sequence. locals_: names that shadow fields (params)."""
    kept = []
    carry = ""
    for t in toks:
        if t.t == "final":
            carry += t.ws
            continue
        if carry:
            t = Tok(t.k, t.t, carry + t.ws if "\n" in carry else t.ws, t.line)
            carry = ""
        kept.append(t)
    toks = rewrite_finally(kept)
    known = ctx.known
    out = []
    locals_ = set(locals_)
    i = 0
    n = len(toks)

    def prev_t():
        return out[-1].t if out else ""

    while i < n:
        t = toks[i]
        nx = toks[i + 1].t if i + 1 < n else ""
        # ---- X.class -------------------------------------------------------------------------------------------
        if t.k == "id" and nx == "." and i + 2 < n and toks[i + 2].t == "class":
            out.append(Tok("id", t.t + "_class", t.ws, t.line))
            i += 3
            continue
        # ---- lambdas -------------------------------------------------------------------------------------------
        if t.t == "->":
            # parameters are already in `out`: either `( a , b )` or a single identifier
            if prev_t() == ")":
                k = len(out) - 1
                depth = 0
                while True:
                    if out[k].t == ")":
                        depth += 1
                    elif out[k].t == "(":
                        depth -= 1
                        if depth == 0:
                            break
                    k -= 1
                names = [x.t for x in out[k + 1:-1] if x.k == "id"]
                ws = out[k].ws
                del out[k:]
            else:
                names = [out[-1].t]
                ws = out[-1].ws
                del out[-1:]
            locals_.update(names)
            head = "[=](%s) mutable" % ", ".join("auto " + x for x in names)
            out.append(Tok("op", head, ws, t.line))
            if nx == "{":
                i += 1
                continue
            # expression body: up to the `,` `)` `;` at depth 0
            j = i + 1
            depth = 0
            while True:
                if toks[j].t in "([{":
                    depth += 1
                elif toks[j].t in ")]}":
                    if depth == 0:
                        break
                    depth -= 1
                elif toks[j].t in (",", ";") and depth == 0:
                    break
                j += 1
            inner = rewrite_body(toks[i + 1:j], ctx, locals_)
            out.append(T("op", "{"))
            out.append(T("id", "return"))
            out.extend(inner)
            out.append(T("op", ";", ""))
            out.append(T("op", "}"))
            i = j
            continue
        # ---- catch (X e) -> catch (X& e) -----------------------------------------------------------------------
        if t.t == "catch" and nx == "(":
            out.append(t)
            out.append(toks[i + 1])
            out.append(Tok("id", toks[i + 2].t + " &", toks[i + 2].ws, t.line))
            i += 3
            continue
        # ---- throw new X(...) -> throw X(...) -----------------------------------------------------------------
        if t.t == "throw" and nx == "new":
            close = match_close(toks, i + 3, "(", ")")
            inner = []
            for arg in split_params(toks[i + 4:close]):        # "text" + value + ...: Java string concatenation -> jconcat(...)
                parts, cur, depth = [], [], 0
                for tk in arg:
                    if tk.t in "([{":
                        depth += 1
                    elif tk.t in ")]}":
                        depth -= 1
                    if tk.t == "+" and depth == 0:
                        parts.append(cur)
                        cur = []
                    else:
                        cur.append(tk)
                parts.append(cur)
                if inner:
                    inner.append(T("op", ",", ""))
                if len(parts) > 1 and any(tk.k == "str" for tk in arg):
                    inner.append(T("id", "jconcat("))
                    for n_, part in enumerate(parts):
                        if n_:
                            inner.append(T("op", ",", ""))
                        inner.extend(rewrite_body(part, ctx, locals_))
                    inner.append(T("op", ")", ""))
                else:
                    inner.extend(rewrite_body(arg, ctx, locals_))
            out.append(t)
            out.append(T("id", "jat(%s(" % toks[i + 2].t))
            out.extend(inner)
            out.append(T("op", '), "%s:%d")' % (ctx.where, t.line), ""))
            i = close + 1
            continue
        # ---- new ----------------------------------------------------------------------------------------------
        if t.t == "new":
            r = parse_type(toks, i + 1, known)
            if r is None:
                # array creation: new T[expr] / new T[]{...}
                r2 = parse_type(toks, i + 1, known | set(PRIMS)) if toks[i + 1].t in PRIMS else None
                if r2 is None:
                    die("cannot translate `new %s` at line %d" % (toks[i + 1].t, t.line))
                r = r2
            cpp, bare, j = r
            if cpp.startswith("JArr<") and toks[j].t == "{":          # new T[]{...}
                out.append(Tok("id", cpp, t.ws, t.line))
                i = j
                continue
            if toks[j].t == "[":                                       # new T[expr]
                close = match_close(toks, j, "[", "]")
                inner = rewrite_body(toks[j + 1:close], ctx, locals_)
                out.append(Tok("id", "JArr<%s>::make" % cpp, t.ws, t.line))
                out.append(T("op", "(", ""))
                out.extend(inner)
                out.append(T("op", ")", ""))
                i = close + 1
                continue
            if bare.endswith("<>"):                                    # diamond: element types from the declaration
                decl = None
                for k in range(len(out) - 1, -1, -1):
                    if out[k].t in (";", "{", "}"):
                        break
                    m = re.match(r"Ref<\w+(<.*>)>$", out[k].t)
                    if m:
                        decl = m.group(1)
                        break
                if decl is None:
                    die("diamond without a declared type at line %d" % t.line)
                bare = bare[:-2] + decl
            out.append(Tok("id", "jnew<%s>" % bare, t.ws, t.line))
            i = j
            continue
        # ---- instanceof ---------------------------------------------------------------------------------------
        if t.t == "instanceof":
            left = out.pop()
            out.append(Tok("id", "jinstanceof(%s, %s_class)" % (left.t, nx), left.ws, t.line))
            i += 2
            continue
        # ---- casts: ( Type ) operand ---------------------------------------------------------------------------
        if t.t == "(" and i + 2 < n:
            r = parse_type(toks, i + 1, known | set(PRIMS))
            if r is not None and toks[r[2]].t == ")" and (toks[r[2] + 1].k in ("id", "num") or toks[r[2] + 1].t == "(") \
                    and prev_t() not in ("if", "while", "for", "switch", "catch") and (not out or out[-1].k != "id" or out[-1].t in ("return",)):
                cpp, bare, j = r
                if bare in PRIMS.values():
                    out.append(Tok("op", "(%s)" % cpp, t.ws, t.line))
                    i = j + 1
                    continue
                # reference cast: operand must be a single identifier (all uses in the translated ranges are)
                op = toks[j + 1]
                if op.k != "id":
                    die("cast operand too complex at line %d" % t.line)
                out.append(Tok("id", "jcast<%s>(%s)" % (bare, op.t), t.ws, t.line))
                i = j + 2
                continue
        # ---- >>> ----------------------------------------------------------------------------------------------
        if t.t == ">>>":
            left = out.pop()
            right = toks[i + 1]
            out.append(Tok("id", "jushr(%s, %s)" % (left.t, right.t), left.ws, t.line))
            i += 2
            continue
        # ---- declarations / for-each: Type name ------------------------------------------------------------------
        if t.k == "id" and prev_t() not in (".", "->", "::", "new") and (t.t in known or t.t in PRIMS or t.t in QUALIFIER_ONLY):
            r = parse_type(toks, i, known)
            if r is not None and r[2] < n and toks[r[2]].k == "id" and toks[r[2]].t not in ("instanceof",):
                cpp, bare, j = r
                out.append(Tok("id", cpp, t.ws, t.line))
                locals_.add(toks[j].t)
                # further declarators of the same statement: `long a = .., b = .., c;`
                k = j
                depth = 0
                while k < n and not (toks[k].t in (";", ":") and depth == 0) and not (toks[k].t == ")" and depth == 0):
                    if toks[k].t in "([{":
                        depth += 1
                    elif toks[k].t in ")]}":
                        depth -= 1
                    elif toks[k].t == "," and depth == 0 and toks[k + 1].k == "id" and toks[k + 2].t in ("=", ",", ";"):
                        locals_.add(toks[k + 1].t)
                    k += 1
                i = j
                continue
            if r is not None and t.t in QUALIFIER_ONLY:
                pass
        # ---- static scopes and qualifier-only names ----------------------------------------------------------------
        if t.k == "id" and nx == "." and prev_t() not in (".", "->") and t.t not in locals_:
            if t.t in QUALIFIER_ONLY and i + 2 < n and toks[i + 2].t in known:
                i += 2                                             # RaftLog.Entry -> Entry
                continue
            if t.t in ctx.static_scopes:
                nm = t.t + "_" if (t.t == "String" or t.t in GENERIC_TYPES) else t.t     # raw generic / String statics live in X_
                # AtomicLongFieldUpdater.newUpdater(X.class, "f") -> AtomicLongFieldUpdater_newUpdater(&X::f)
                if t.t in ("AtomicLongFieldUpdater", "AtomicIntegerFieldUpdater") and toks[i + 2].t == "newUpdater":
                    close = match_close(toks, i + 3, "(", ")")
                    clsn, fld = toks[i + 4].t, toks[close - 1].t.strip('"')
                    out.append(Tok("id", "%s_newUpdater(&%s::%s)" % (t.t, clsn, fld), t.ws, t.line))
                    i = close + 1
                    continue
                out.append(Tok("id", nm, t.ws, t.line))
                out.append(Tok("op", "::", "", t.line))
                i += 2
                continue
        # ---- member access ----------------------------------------------------------------------------------------
        if t.t == ".":
            out.append(Tok("op", "->", t.ws, t.line))
            # field whose name collides with a method somewhere: x.term -> x->term_
            f = toks[i + 1]
            after = toks[i + 2].t if i + 2 < n else ""
            if f.k == "id" and after != "(" and (f.t in ctx.collide_any):
                out.append(Tok("id", f.t + "_", f.ws, f.line))
                i += 2
                continue
            i += 1
            continue
        # ---- bare identifiers -------------------------------------------------------------------------------------
        if t.k == "id":
            if t.t == "null":
                out.append(Tok("id", "nullptr", t.ws, t.line))
            elif t.t in PRIMS and nx != ".":
                out.append(Tok("id", PRIMS[t.t], t.ws, t.line))
            elif t.t in ctx.collide_here and nx != "(" and prev_t() not in (".", "->", "::") and t.t not in locals_:
                out.append(Tok("id", t.t + "_", t.ws, t.line))
            elif nx == "(" and t.t in locals_ and prev_t() not in (".", "->", "::"):
                out.append(Tok("id", "this->" + t.t, t.ws, t.line))      # a local shadows the method of the same name
            else:
                out.append(t)
            i += 1
            continue
        if t.k == "num":
            s = t.t.replace("_", "")
            if s[-1] in "lL":
                s = "(jlong)" + s[:-1] + "LL"
            out.append(Tok("num", s, t.ws, t.line))
            i += 1
            continue
        out.append(t)
        i += 1
    return out


# ------------------------------------------------------------------------------------------------------------------
# emission

class Unit:
    """everything produced for the generated headers"""

    def __init__(self):
        self.fwd, self.decls, self.defs, self.incs = [], [], [], {}


def field_names(node):
    names = []
    for m in node.members:
        if isinstance(m, Field):
            toks = m.toks
            depth = 0
            for k, tk in enumerate(toks):
                if tk.t in "([{<":
                    depth += 1
                elif tk.t in ")]}>":
                    depth -= 1
                elif depth == 0 and tk.k == "id" and toks[k + 1].t in ("=", ",", ";") and k > 0 and toks[k - 1].t not in (".", "="):
                    # declarator names: identifier directly before = , ;  (outside any initializer expression)
                    names.append(tk.t)
            # initializer expressions may contain `x = y` only inside brackets, which depth excludes
    return names


def declared_names_of_field(toks):
    """declarator names of one field statement (first is after the type)"""
    names, depth, in_init = [], 0, False
    for k, tk in enumerate(toks):
        if tk.t in "([{":
            depth += 1
        elif tk.t in ")]}":
            depth -= 1
        elif depth == 0 and tk.t == "=":
            in_init = True
        elif depth == 0 and tk.t in (",", ";"):
            in_init = False
        elif depth == 0 and not in_init and tk.k == "id" and toks[k + 1].t in ("=", ",", ";"):
            names.append(tk.t)
    return names


def emit_class(node, unit, all_nodes, known, static_scopes, collide_any, entry, outer=None, statics_base=None):
    name = node.name
    hoist = None
    if outer is None and entry.get("hoist_statics"):
        # a nested Java class sees the static fields of its outer class unqualified; nested classes are flattened to top
        # level here, so the outer class's statics move into <Outer>_statics, which the outer class and its nested classes inherit
        hoist = []
        statics_base = name + "_statics"
        unit.fwd.append("struct %s;" % statics_base)
        unit.decls.append(None)                          # placeholder, filled once the statics are known
        hoist_slot = len(unit.decls) - 1
    # flatten nested types first (they become top-level structs declared BEFORE the outer one)
    keep = entry.get("keep")
    for m in node.members:
        if isinstance(m, ClassNode) and (keep is None or any(a <= m.line <= b for a, b in keep)):
            emit_class(m, unit, all_nodes, known, static_scopes, collide_any, entry, outer=node, statics_base=statics_base)
    exclude = set(entry.get("exclude", []))

    def kept(m):
        if isinstance(m, Method) and m.name in exclude:
            return False
        if keep is None:
            return True
        return any(a <= m.line <= b for a, b in keep)

    members = [m for m in node.members if not isinstance(m, ClassNode) and kept(m)
               and not (isinstance(m, Field) and any(t.t == "LoggerFactory" for t in m.toks))]   # slf4j loggers: one global no-op
    bases = []
    for b in node.extends:
        if b not in DROP_BASES:
            bases.append("public " + b)
    for b in node.implements:
        if b not in DROP_BASES:
            bases.append("public virtual " + b)
    if not node.extends or all(b in DROP_BASES for b in node.extends):
        bases.insert(0, "public virtual Object")
    if statics_base:
        bases.append("public " + statics_base)
    collide_here = all_nodes[name]["collide"]
    ctx = Ctx(name, collide_here, collide_any, known, static_scopes, os.path.basename(entry["file"]))
    is_iface = node.kind == "interface"

    unit.fwd.append("struct %s; extern Ref<Class> %s_class;" % (name, name))
    d = ["// %s:%d" % (entry["file"], node.line), "struct %s : %s {" % (name, ", ".join(bases))]
    d.append("    Class *klass_() override { return %s_class.get(); }" % name)
    for m in members:
        if isinstance(m, Field):
            toks = [t for t in m.toks if t.t not in DROP_MODIFIERS and t.t != "static"]
            r = parse_type(toks, 0, known)
            if r is None:
                die("cannot parse field at %s:%d: %s" % (entry["file"], m.line, render(toks)))
            cpp, bare, j = r
            rest = toks[j:]
            # rename colliding declarators, zero-initialise the ones without initializer
            names = declared_names_of_field(rest)
            body = rewrite_body(rest, Ctx(name, set(), collide_any, known, static_scopes), set(names))
            txt = []
            for k, tk in enumerate(body):
                s = tk.t
                if tk.k == "id" and s in names and body[k + 1].t in ("=", ",", ";"):
                    if s in collide_here:
                        s = s + "_"
                    if body[k + 1].t in (",", ";"):
                        s = s + "{}"
                txt.append(tk.ws + s)
            static = m.static or is_iface
            const = static and cpp in PRIMS.values() and any(tk.t == "=" for tk in body)
            line_ = "    %s%s%s  // :%d" % ("static constexpr " if const else "static inline " if static else "", cpp, "".join(txt), m.line)
            if static and hoist is not None:
                hoist.append(line_)
            else:
                d.append(line_)
        else:
            ptxt, pnames = cpp_params(m.params, known)
            if m.is_ctor:
                d.append("    %s(%s);  // :%d" % (name, ptxt, m.line))
                head = "inline %s::%s(%s)" % (name, name, ptxt)
                body = m.body
                init = ""
                # super(...) / this(...) as first statement -> initializer list
                if len(body) > 2 and body[1].t in ("super", "this") and body[2].t == "(":
                    close = match_close(body, 2, "(", ")")
                    args = rewrite_body(body[3:close], ctx, pnames)
                    target = (node.extends[0] if body[1].t == "super" else name)
                    init = " : %s(%s)" % (target, render(args).strip())
                    body = [body[0]] + body[close + 2:]
                unit.defs.append("// %s:%d\n%s%s%s\n" % (entry["file"], m.line, head, init, render(rewrite_body(body, ctx, pnames))))
            else:
                r = parse_type(m.ret, 0, known)
                if r is None or r[2] != len(m.ret):
                    die("cannot parse return type of %s.%s (%s:%d)" % (name, m.name, entry["file"], m.line))
                ret = r[0]
                if m.static:
                    pre = "static "
                else:
                    pre = "virtual "
                if not m.has_body:
                    d.append("    %s%s %s(%s) = 0;  // :%d" % (pre, ret, m.name, ptxt, m.line))
                else:
                    d.append("    %s%s %s(%s);  // :%d" % (pre, ret, m.name, ptxt, m.line))
                    unit.defs.append("// %s:%d\ninline %s %s::%s(%s)%s\n" % (
                        entry["file"], m.line, ret, name, m.name, ptxt, render(rewrite_body(m.body, ctx, pnames))))
    for lift in entry.get("lift_done", []) if outer is None else []:
        d.append("    " + lift["decl"])
        unit.defs.append(lift["def"])
    if outer is None and entry.get("extra_members"):
        d.append(entry["extra_members"].rstrip("\n"))
    d.append("};")
    if hoist is not None:
        unit.decls[hoist_slot] = "struct %s {\n%s\n};\n" % (statics_base, "\n".join(hoist))
    unit.decls.append("\n".join(d) + "\n")
    if outer is None and entry.get("extra_defs"):
        unit.defs.append(entry["extra_defs"])


def collect(node, table, entry):
    fields, methods = [], []
    for m in node.members:
        if isinstance(m, Field):
            toks = [t for t in m.toks if t.t not in DROP_MODIFIERS and t.t != "static"]
            r = parse_type(toks, 0, set(table["known"]))
            if r is not None:
                fields += declared_names_of_field(toks[r[2]:])
        elif isinstance(m, Method):
            methods.append(m.name)
        else:
            collect(m, table, entry)
    table["nodes"][node.name] = dict(fields=fields, methods=methods, extends=node.extends, collide=set())


def names_in(node):
    out = [node.name]
    for m in node.members:
        if isinstance(m, ClassNode):
            out += names_in(m)
    return out


def read_lines(path):
    with open(path, encoding="utf-8") as f:
        return f.read().split("\n")


def sha_of(lines, a=None, b=None):
    text = "\n".join(lines if a is None else lines[a - 1:b])
    return hashlib.sha256(text.encode("utf-8")).hexdigest()          # full-length sha256 (round 2 stored the first 32 hex digits)


def apply_substitutions(rel, lines):
    """returns {line_no: replacement} and blanks the replaced lines"""
    subs = {}
    for f, a, b, repl in SUBSTITUTIONS:
        if f == rel:
            for k in range(a, b + 1):
                lines[k - 1] = ""
            lines[a - 1] = "__SUBST_%d__;" % a
            subs["__SUBST_%d__" % a] = repl
    return subs


def finish_subst(text, subs):
    for k, v in subs.items():
        text = text.replace(k + ";", v)
    return text


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "oracle", "_ref", "gen"))
    ap.add_argument("--print-sha", action="store_true")
    args = ap.parse_args()
    base = os.path.join(args.ref, SRC)
    if not os.path.isdir(base):
        die("reference checkout not found at %s" % base)
    pins_path = os.path.join(ROOT, "oracle", "ref_shim", "source_pins.json")
    import json
    pins = json.load(open(pins_path)) if os.path.exists(pins_path) and not args.print_sha else {}
    new_pins = {}

    # pass 1: parse everything, learn the type names and the field/method collisions
    parsed = []
    table = dict(nodes={}, known=set(KNOWN_TYPES))
    for e in MANIFEST:
        lines = read_lines(os.path.join(base, e["file"]))
        subs = apply_substitutions(e["file"], lines)
        e["_subs"] = subs
        if e["mode"] == "class":
            key = e["file"]
            new_pins[key] = sha_of(read_lines(os.path.join(base, e["file"])))
            for lf in e.get("lift", []):
                new_pins["%s:%d-%d" % (e["file"], lf["lines"][0], lf["lines"][1])] = sha_of(
                    read_lines(os.path.join(base, e["file"])), lf["lines"][0], lf["lines"][1])
            toks = tokenize("\n".join(lines))
            i = 0
            while toks[i].t not in ("class", "interface"):
                i += 1
            node, _ = parse_class(toks, i)
            table["known"].update(names_in(node))
            parsed.append((e, node, toks))
        else:
            orig = read_lines(os.path.join(base, e["file"]))
            for a, b in e["ranges"]:
                new_pins["%s:%d-%d" % (e["file"], a, b)] = sha_of(orig, a, b)
            table["known"].add(e["cls"])
            parsed.append((e, None, lines))
    if args.print_sha:
        json.dump(new_pins, open(pins_path, "w"), indent=1, sort_keys=True)
        print("wrote %s (%d pins)" % (pins_path, len(new_pins)))
        return
    for k, v in new_pins.items():
        if pins.get(k) != v:
            die("reference source changed: %s (sha %s, pinned %s). Re-read the range, then run --print-sha." % (k, v, pins.get(k)))
    known = table["known"]
    for e, node, _ in parsed:
        if node is not None:
            collect(node, table, e)
    for e, node, _ in parsed:
        if node is None:
            table["nodes"][e["cls"]] = dict(fields=list(e.get("collisions", [])), methods=list(e.get("collisions", [])),
                                            extends=[], collide=set(e.get("collisions", [])))
    nodes = table["nodes"]
    for nm, inf in nodes.items():
        inf["collide"] = set(inf["fields"]) & set(inf["methods"])
    changed = True
    while changed:                       # inherited collisions
        changed = False
        for nm, inf in nodes.items():
            for b in inf["extends"]:
                if b in nodes and not nodes[b]["collide"] <= inf["collide"]:
                    inf["collide"] |= nodes[b]["collide"]
                    changed = True
    collide_any = set()
    for inf in nodes.values():
        collide_any |= inf["collide"]
    static_scopes = set(STATIC_SCOPES) | set(nodes.keys()) | {"AtomicLongFieldUpdater", "AtomicIntegerFieldUpdater"}

    units = {}
    for e, node, payload in parsed:
        unit = units.setdefault(e.get("unit", "ref"), Unit())
        if e["mode"] == "class":
            # lifted lambda bodies
            e["lift_done"] = []
            lines = read_lines(os.path.join(base, e["file"]))
            for lf in e.get("lift", []):
                a, b = lf["lines"]
                new_key = "%s:%d-%d" % (e["file"], a, b)
                if pins.get(new_key) != sha_of(lines, a, b):
                    die("reference source changed: %s" % new_key)
                body_lines = lines[a - 1:b]
                # first line: `response.on(head, timeout, (result, error, canceled) -> {`  last line: `});`
                first = body_lines[0]
                if "->" not in first or not body_lines[-1].strip().startswith("})"):
                    die("lift range %s does not look like a callback lambda" % new_key)
                inner = "{\n" + "\n".join(body_lines[1:-1]) + "\n}"
                btoks = tokenize(inner, a)
                ptoks = tokenize(lf["params"])
                ptxt, pnames = cpp_params(ptoks, known)
                ctx = Ctx(node.name, nodes[node.name]["collide"], collide_any, known, static_scopes, os.path.basename(e["file"]))
                e["lift_done"].append({
                    "decl": "virtual void %s(%s);  // lifted lambda body :%d-%d" % (lf["name"], ptxt, a, b),
                    "def": "// %s:%d-%d (lambda body lifted into a method; captured variables are parameters)\ninline void %s::%s(%s)%s\n" % (
                        e["file"], a, b, node.name, lf["name"], ptxt, render(rewrite_body(btoks, ctx, pnames)))})
            emit_class(node, unit, nodes, known, static_scopes, collide_any, e)
            unit.defs[:] = [finish_subst(x, e["_subs"]) for x in unit.defs]
        else:
            cls = e["cls"]
            ctx = Ctx(cls, nodes[cls]["collide"], collide_any, known, static_scopes, os.path.basename(e["file"]))
            lines = payload
            decls = []
            for a, b in e["ranges"]:
                text = "\n".join(lines[a - 1:b])
                toks = tokenize(text, a)
                members = parse_members(toks, 0, len(toks), cls)
                for m in members:
                    if not isinstance(m, Method) or not m.has_body:
                        die("%s:%d-%d is not a sequence of methods" % (e["file"], a, b))
                    ptxt, pnames = cpp_params(m.params, known)
                    r = parse_type(m.ret, 0, known)
                    if r is None:
                        die("cannot parse return type of %s.%s" % (cls, m.name))
                    decls.append("    virtual %s %s(%s);  // %s:%d" % (r[0], m.name, ptxt, e["file"], m.line))
                    unit.defs.append(finish_subst("// %s:%d\ninline %s %s::%s(%s)%s\n" % (
                        e["file"], m.line, r[0], cls, m.name, ptxt, render(rewrite_body(m.body, ctx, pnames))), e["_subs"]))
            unit.incs[cls] = "\n".join(decls) + "\n"

    os.makedirs(args.out, exist_ok=True)
    banner = "// GENERATED by tools/make_ref.py from the reference's Java sources — do not edit, do not commit.\n"
    for uname, unit in units.items():
        with open(os.path.join(args.out, "%s_fwd.hpp" % uname), "w") as f:
            f.write(banner + "\n".join(dict.fromkeys(unit.fwd)) + "\n")
        with open(os.path.join(args.out, "%s_decls.hpp" % uname), "w") as f:
            f.write(banner + "\n".join(unit.decls))
        with open(os.path.join(args.out, "%s_defs.hpp" % uname), "w") as f:
            f.write(banner + "\n".join(unit.defs))
        for cls, text in unit.incs.items():
            with open(os.path.join(args.out, "%s.decls.inc" % cls), "w") as f:
                f.write(banner + text)
        print("make_ref: unit %s: %d classes, %d definitions -> %s" % (uname, len(unit.decls), len(unit.defs), args.out))


if __name__ == "__main__":
    main()
