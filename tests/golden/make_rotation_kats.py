#!/usr/bin/env python3
"""Extracts the known answers of the reference's rotation tests into tests/golden/rotation_kats.json.

Reads /root/reference/test/src/rot2rpy.cpp and rpy2rot.cpp as text and keeps only data: per SCENARIO its title,
the numeric inputs (`double x = ...;`, `M << ...;`) and every `CHECK(M(i,j) == Approx(value))`.  No code of the
reference is kept.  Run here (the reference tree does not travel to the GPU box): python tests/golden/make_rotation_kats.py
"""
import json
import math
import os
import re

REF = "/root/reference/test/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rotation_kats.json")


def number(tok):
    return float(eval(tok.replace("M_PI", repr(math.pi)), {"__builtins__": {}}, {}))


def scenarios(path):
    txt = open(path).read()
    parts = re.split(r'SCENARIO\("([^"]*)"\)', txt)[1:]
    for title, body in zip(parts[0::2], parts[1::2]):
        case = {"title": title, "scalars": {}, "matrices": {}, "checks": {}}
        for m in re.finditer(r"double\s+(\w+)\s*=\s*([^;]+);", body):
            case["scalars"][m.group(1)] = number(m.group(2).strip())
        for m in re.finditer(r"(\w+)\s*<<([^;]+);", body):
            case["matrices"][m.group(1)] = [number(t.strip()) for t in m.group(2).split(",")]      # row-major, as written
        for m in re.finditer(r"CHECK\((\w+)\((\d+),(\d+)\)\s*==\s*Approx\(\s*([-+0-9.eE]+)\s*\)\)", body):
            case["checks"].setdefault(m.group(1), []).append([int(m.group(2)), int(m.group(3)), number(m.group(4))])
        yield case


if __name__ == "__main__":
    out = {"source": "test/src/rot2rpy.cpp, test/src/rpy2rot.cpp of the reference (known answers only)",
           "rot2rpy": list(scenarios(os.path.join(REF, "rot2rpy.cpp"))),
           "rpy2rot": list(scenarios(os.path.join(REF, "rpy2rot.cpp")))}
    with open(OUT, "w") as fh:
        json.dump(out, fh, indent=1)
    print("%d rot2rpy scenarios, %d rpy2rot scenarios -> %s" % (len(out["rot2rpy"]), len(out["rpy2rot"]), OUT))
