"""CPU: every `asr_*(...)` call written in INTEGRATION.md has the argument count of its prototype in include/asr_hip.h
(VERDICT r4 Weak #9: the document showed asr_train_step with an in_mode argument that does not exist and asr_topk
without ctx - a maintainer following it got a crash, not an error code)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _split_args(text):
    """top-level comma split of the text between a call's parentheses"""
    args, depth, cur = [], 0, ""
    for ch in text:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            args.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        args.append(cur.strip())
    return args


def _calls(text, names):
    """(name, argument list) of every call `name(` in text whose parentheses close"""
    out = []
    for m in re.finditer(r"\b(asr_[a-z0-9_]+)\s*\(", text):
        name = m.group(1)
        if name not in names:
            continue
        i, depth = m.end(), 1
        while i < len(text) and depth:
            depth += text[i] in "([{"
            depth -= text[i] in ")]}"
            i += 1
        if depth:
            continue
        out.append((name, _split_args(text[m.end():i - 1]), text.count("\n", 0, m.start()) + 1))
    return out


def header_prototypes():
    src = open(os.path.join(ROOT, "include", "asr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int|void|const char \*)\s*(asr_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        args = _split_args(m.group(2))
        protos[m.group(1)] = 0 if args == ["void"] else len(args)
    return protos


def test_every_call_in_integration_md_matches_the_header():
    protos = header_prototypes()
    assert len(protos) > 60 and protos["asr_train_step"] == 7 and protos["asr_topk"] == 12
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    calls = _calls(doc, protos)
    assert len(calls) >= 20
    bad = []
    for name, args, line in calls:
        if any(a.strip() in ("...", "…") or a.strip().endswith("...") for a in args):
            continue                                   # an elided argument list is prose, not a call to copy
        if len(args) != protos[name]:
            bad.append("INTEGRATION.md:%d %s has %d arguments, the header declares %d" % (line, name, len(args), protos[name]))
    assert not bad, "\n".join(bad)
    # every function the document names exists
    unknown = sorted(set(re.findall(r"\b(asr_[a-z0-9_]+)\s*\(", doc)) - set(protos) -
                     {"asr_fused_prepare"})
    assert not unknown, unknown


def test_the_stub_in_integration_md_declares_the_header_struct():
    """the ctypes structure the document shows a maintainer has the members of asr_config, in order (a stub written
    against the 64-byte struct of round 4 still loads - asr_create accepts that size - but the document shows today's)"""
    header = open(os.path.join(ROOT, "include", "asr_hip.h")).read()
    body = re.search(r"typedef struct asr_config \{(.*?)\} asr_config;", header, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    members = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            members += [n.strip() for n in decl.split(None, 1)[1].split(",")]
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = doc[doc.index("class _Cfg(ctypes.Structure)"):doc.index("def _chk(ctx, rc)")]
    assert re.findall(r'"([a-zA-Z0-9_]+)"', stub) == members
    # ... and constructs it with one value per member
    ctor = re.search(r"cfg = _Cfg\((.*?)\)\n", doc, flags=re.S).group(1)
    assert len(_split_args(ctor)) == len(members)
