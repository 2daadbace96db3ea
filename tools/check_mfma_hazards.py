"""Static check of the Hamming matrix-core kernel's ISA: how many wait states lie between an
MFMA and the first vector instruction that reads its result.

gfx950 does not interlock that read: the compiler inserts the wait states an instruction it
selected itself needs, but none for a reader inside an asm statement, and it may move such a
statement up to right behind the MFMA.  hamming_mfma.hip keeps the fold of an accumulator
several MFMAs behind its last writer; this script compiles the file to assembly and measures
the distance in the code the compiler actually emitted: one wait state per instruction, N + 1
per `s_nop N`, eight per MFMA in between (an 8-pass MFMA holds the pipe that long).  An 8-pass
MFMA needs 11.  The fold is written with compiler-visible instructions today; the check stays
as a guard for the next rewrite.

    python3 tools/check_mfma_hazards.py [--min 11]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "sfm_amd", "csrc", "hamming_mfma.hip")


def compile_to_asm(path):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
           "-fno-fast-math", "-fno-honor-nans", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(SRC), "-S",
           "--cuda-device-only", SRC, "-o", path]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def kernels(asm):
    for name in re.findall(r"^(\w*hamming_mfma_kernel\w+):", asm, re.M):
        start = asm.index(name + ":")
        yield name, asm[start:asm.index(".end_amdhsa_kernel", start)]


def registers(text):
    regs = [int(x) for x in re.findall(r"\bv(\d+)\b", text)]
    for a, b in re.findall(r"v\[(\d+):(\d+)\]", text):
        regs += range(int(a), int(b) + 1)
    return regs


def min_distance(body):
    """Smallest number of wait states between an MFMA and a vector instruction reading one of
    the registers it wrote, walking the kernel in layout order (loops are entered from above,
    so the first trip sees the distances of every later one or shorter ones)."""
    written = {}
    now, worst = 0, None
    for raw in body.split("\n"):
        line = raw.split(";")[0].strip()
        if not line or line.startswith("."):
            continue
        op = line.split()[0]
        if op.startswith("v_mfma"):
            now += 8
            dst = re.match(r"\S+ v\[(\d+):(\d+)\]", line)
            for r in range(int(dst.group(1)), int(dst.group(2)) + 1):
                written[r] = now
            continue
        now += int(line.split()[1]) + 1 if op == "s_nop" else 1
        if not op.startswith("v_"):
            continue
        operands = line[len(op):].split(",")
        sources = operands if op.startswith("v_cmp") else operands[1:]
        for r in registers(",".join(sources)):
            if r in written:
                d = now - written[r]
                if worst is None or d < worst[0]:
                    worst = (d, line)
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--min", type=int, default=11)
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "hamming_mfma.s")
        compile_to_asm(path)
        asm = open(path).read()
    ok = True
    for name, body in kernels(asm):
        worst = min_distance(body)
        print(name, worst)
        if worst is None or worst[0] < args.min:
            ok = False
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
