#!/usr/bin/env python3
"""Extracts the known-answer cases of the reference's candidate-finder tests into tests/golden/finder_cases.json.

Reads  <reference>/src/test/Pisces.Domain.Tests/UnitTests/Logic/VariantFinderTests.cs  (SnvTests :118-194, MnvTests :196-485, DeletionTests :487-747 and
InsertionTests :749-1020): each ExecuteTest(new CandidateVariantsTest(start, refUnderRead, cigar, readBases, qualities)
{ Expectations = ... }) call becomes one JSON case (inputs + expected candidates).  Only DATA is kept: the statements of the
two test bodies are evaluated by a tiny interpreter (string / int assignments, QualitiesArray(...), the expectation
constructors) and nothing of the test code is stored.  Coordinates are stored relative to the read start.
Run from the repo root in the build container (the reference is not available on the GPU box):
    python tests/golden/extract_finder_cases.py /root/reference
"""
import copy
import json
import os
import re
import sys

ref_root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
src = os.path.join(ref_root, "src/test/Pisces.Domain.Tests/UnitTests/Logic/VariantFinderTests.cs")
text = open(src, encoding="utf-8-sig").read()


def body_of(name):
    i = text.index("public void %s()" % name)
    i = text.index("{", i)
    depth, j = 0, i
    while True:
        c = text[j]
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                return text[i + 1:j], text.count("\n", 0, i) + 1
        j += 1


def statements(body):
    body = re.sub(r"//[^\n]*", "", body)
    out, depth, cur, in_str = [], 0, [], False
    for ch in body:
        if ch == '"':
            in_str = not in_str
        if not in_str:
            if ch in "({[":
                depth += 1
            elif ch in ")}]":
                depth -= 1
            elif ch == ";" and depth == 0:
                out.append("".join(cur).strip())
                cur = []
                continue
        cur.append(ch)
    return [s for s in out if s]


def read_span(cigar):
    return sum(int(n) for n, op in re.findall(r"(\d+)([MIDNSHP=X])", cigar) if op in "MIS=X")


def QualitiesArray(cigar, primary, substitute=None, sites=None):
    q = [primary] * read_span(cigar)
    if substitute is not None and sites is not None:
        for s in sites:
            q[s] = substitute
    return q


def EV(type_, ref, alt, coord, open_left=None, open_right=None):
    return {"type": type_, "ref": ref, "alt": alt, "coord": coord, "open_left": open_left, "open_right": open_right}


def Expect(*a):
    if not a:
        return {"n": 0, "variants": []}
    if isinstance(a[1], list):
        return {"n": a[0], "variants": a[1]}
    return {"n": a[0], "variants": [EV(*a[1:])]}


def convert(expr):
    e = expr
    e = e.replace("CandidateFinderTestHelpers.QualitiesArray(", "QualitiesArray(")
    e = re.sub(r"new\s+List<ExpectedVariant>\s*\{", "[", e)
    e = re.sub(r"new\s*\[\]\s*\{", "[", e)
    e = re.sub(r"CandidateVariantType\.(\w+)", r'"\1"', e)
    e = e.replace("new ExpectedVariant(", "EV(").replace("new CandidateVariantTestExpectations(", "Expect(")
    e = re.sub(r"\btrue\b", "True", e)
    e = re.sub(r"\bfalse\b", "False", e)
    # the remaining braces close the lists opened above
    return e.replace("}", "]")


cases = []
for fn in ("SnvTests", "MnvTests", "DeletionTests", "InsertionTests"):
    body, line0 = body_of(fn)
    env = {"_readStartPos": 1234567, "_qualityCutoff": 20, "QualitiesArray": QualitiesArray, "EV": EV, "Expect": Expect}
    for st in statements(body):
        m = re.match(r"ExecuteTest\(\s*new CandidateVariantsTest\((.*?)\)\s*(?:\{\s*Expectations\s*=\s*(.*)\}\s*)?\)$", st, re.S)
        if m:
            args = eval("[" + " ".join(convert(m.group(1)).split()) + "]", env)
            exp = eval(" ".join(convert(m.group(2)).split()), env) if m.group(2) else None
            start, ref_under_read, cigar, bases, quals = args[:5]
            max_mnv = args[5] if len(args) > 5 else 20
            max_gap = args[6] if len(args) > 6 else 2
            if len(args) > 7:
                exp = args[7]
            exp = copy.deepcopy(exp)
            for v in exp["variants"]:
                v["coord"] = v["coord"] - start
            cases.append({"test": fn, "cigar": cigar, "ref_under_read": ref_under_read, "read": bases, "quals": quals,
                          "max_mnv_length": max_mnv, "max_gap": max_gap, "expected_count": exp["n"], "expected": exp["variants"]})
            continue
        m = re.match(r"(?:(?:var|byte\[\]|string|int)\s+)?(\w+)\s*=\s*(.*)$", st, re.S)
        if m:
            env[m.group(1)] = eval(" ".join(convert(m.group(2)).split()), env)
            continue
        raise SystemExit("unhandled statement: " + st[:120])

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "finder_cases.json")
json.dump({"_source": "Pisces.Domain.Tests/UnitTests/Logic/VariantFinderTests.cs: SnvTests :118-194, MnvTests :196-485, DeletionTests :487-747, InsertionTests :749-1020 (inputs and "
                      "expectations of every ExecuteTest call), harness :28-41 (reference = N x (start - 1 - leading soft clip) + ref_under_read + "
                      "NNNNN), finder settings :1030-1034 (minBQ 20, callMNVs on, maxMnv / maxGap per case); coord is relative to the read start",
           "min_base_call_quality": 20, "call_mnvs": True, "cases": cases}, open(out, "w"), indent=0)
print(out, len(cases), "cases")
