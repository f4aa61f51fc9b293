"""Static check of the hand-written DPP instructions (sp_diag.h: v_fmac_f64_dpp in inline assembly).

The hardware wants two wait states between a VALU write of a register and its read through DPP.  The compiler keeps
that for the instructions it emits, but inline assembly is opaque to its hazard recogniser: it may schedule the
instruction that PRODUCES a DPP operand directly in front of the asm statement (it did: the select that initialises a
column of the pivot block's inverse, in front of that column's first step -- an inverse wrong in its tenth digit).  Every
asm DPP instruction therefore has to bring its own idle states or sit provably behind another one; this script compiles
the kernels that include sp_diag.h to assembly and looks at what is in front of each v_fmac_f64_dpp.

    python tools/check_dpp_hazard.py [extra hipcc flags ...]        exit code 1 and the offending pairs if any"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "starry_process_amd", "csrc")
FILES = ["sp_panel.hip", "sp_small.hip"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-fast-math", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
         "-Wno-unused-function", "--cuda-device-only", "-S"]


def written(instr):
    """(lo, hi) of the vector registers a VALU / LDS / memory instruction writes, or None"""
    m = re.match(r"(v_|ds_read|ds_bpermute|global_load|scratch_load|buffer_load)\S*\s+(v\[(\d+):(\d+)\]|v(\d+))", instr)
    if not m or instr.startswith(("v_cmp", "v_nop")):
        return None
    if m.group(3):
        return int(m.group(3)), int(m.group(4))
    return int(m.group(5)), int(m.group(5))


def scan(asm_path):
    lines = [l.strip() for l in open(asm_path) if l.strip() and not l.strip().startswith((";", ".", "//"))]
    bad, seen = [], 0
    for i, l in enumerate(lines):
        m = re.match(r"v_fmac_f64_dpp v\[(\d+):(\d+)\], v\[(\d+):(\d+)\]", l)
        if not m:
            continue
        seen += 1
        src = (int(m.group(3)), int(m.group(4)))
        states, j = 0, i - 1
        while states < 2 and j >= 0:
            p = lines[j]
            j -= 1
            if p.endswith(":"):          # a label: whatever jumps here is out of this scan's sight -- count it as a hazard
                bad.append((p, l))       # (none today: the chains are straight-line code)
                break
            if p.startswith("s_nop"):
                states += int(p.split()[1], 0) + 1
                continue
            w = written(p)
            if w and not (w[1] < src[0] or w[0] > src[1]):
                bad.append((p, l))
                break
            states += 1
    return seen, bad


def check(extra=()):
    out = {}

    def one(f):
        with tempfile.TemporaryDirectory() as d:
            s = os.path.join(d, f + ".s")
            subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + list(extra) + ["-I", CSRC, os.path.join(CSRC, f), "-o", s],
                           check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            return f, scan(s)

    with ThreadPoolExecutor(len(FILES)) as ex:
        for f, r in ex.map(one, FILES):
            out[f] = r
    return out


if __name__ == "__main__":
    rc = 0
    for f, (seen, bad) in check(sys.argv[1:]).items():
        print("%s: %d v_fmac_f64_dpp, %d with a write of their DPP source less than two wait states ahead" % (f, seen, len(bad)))
        for p, l in bad[:8]:
            print("    %s\n      -> %s" % (p, l[:90]))
        rc |= 1 if bad else 0
    sys.exit(rc)
