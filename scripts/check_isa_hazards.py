#!/usr/bin/env python3
"""check_isa_hazards.py [libturborc_hip.so] -- the gfx940+ rule the compiler enforces on its own code, checked on the SHIPPED ISA:

    a VALU instruction that writes an SGPR / VCC must be followed by two wait states before a VALU instruction reads that
    register (LLVM GCNHazardRecognizer::checkVALUHazards, hasVDecCoExecHazard(); carry-in, v_cndmask's mask, lane-mask and
    constant operands alike).  hipcc pads its own instructions; nothing inside an `asm` string is padded
    (cdna_hip_programming.md 5.7 item 2), so hand-written carry chains have to carry the states themselves.

Walks `llvm-objdump -d` of every kernel in the code object linearly (a label does not reset the window: conservative) and
reports every (writer, reader) pair with fewer than two wait states between them.  Wait states: one per instruction issued in
between, `s_nop N` = N + 1.  Exit status 1 if any site is found.  Used by tests/test_isa_hazards.py (CPU side: the library is
cross-compiled here) and by hand.
"""
import re
import subprocess
import sys
import os

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

SREG = re.compile(r"\b(vcc(?:_lo|_hi)?|s\[(\d+):(\d+)\]|s(\d+))\b")


def regs_of(op):
    """set of scalar register numbers named by an operand string ('vcc' -> {'vcc'}, 's[4:5]' -> {4, 5})"""
    out = set()
    for m in SREG.finditer(op):
        if m.group(1).startswith("vcc"):
            out.add("vcc")
        elif m.group(2) is not None:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add(int(m.group(4)))
    return out


def split_ops(s):
    ops, depth, cur = [], 0, ""
    for ch in s:
        if ch == "[":
            depth += 1
        if ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    return ops


def classify(mn, ops):
    """(scalar registers written, scalar registers read) by a VALU instruction"""
    w, r = set(), set()
    if not mn.startswith("v_"):
        return w, r
    base = mn
    dst_n = 1
    if mn.startswith("v_cmp") or mn.startswith("v_cmpx"):
        if ops:
            w |= regs_of(ops[0])
        for o in ops[1:]:
            r |= regs_of(o)
        if mn.startswith("v_cmpx"):
            pass
        return w, r
    if re.match(r"v_(add|sub|subrev)_co_u32", base) or re.match(r"v_(addc|subb|subbrev)_co_u32", base) or base.startswith("v_mad_u64_u32") or base.startswith("v_mad_i64_i32") or base.startswith("v_div_scale"):
        dst_n = 2
    if base.startswith("v_readfirstlane") or base.startswith("v_readlane"):
        w |= regs_of(ops[0])
        for o in ops[1:]:
            r |= regs_of(o)
        return w, r
    for i, o in enumerate(ops):
        # modifiers (dst_sel:..., quad_perm:[...], row_mask) are not operands
        if ":" in o and not o.startswith("s[") and not o.startswith("v[") and not o.startswith("a["):
            continue
        if i < dst_n:
            if i == 1:
                w |= regs_of(o)
        else:
            r |= regs_of(o)
    if base.startswith("v_div_fmas"):
        r.add("vcc")
    return w, r


def scan(text):
    sites = []
    kernel = None
    window = []                                                # (wait states since, regs written, text)
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            kernel = m.group(1); window = []
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//", line)
        if not m:
            continue
        mn, rest = m.group(1), m.group(2)
        # drop trailing modifiers separated by spaces (e.g. "quad_perm:[1,0,3,2] row_mask:0xf") from the last operand
        rest = re.sub(r"\s+[a-z_0-9]+:(\[[^\]]*\]|\S+)", "", rest)
        ops = split_ops(rest)
        states = 1
        if mn == "s_nop":
            states = int(ops[0], 0) + 1 if ops else 1
        w, r = classify(mn, ops)
        if r:
            for (since, wr, wtext) in window:
                hit = wr & r
                if hit and since < 2:
                    sites.append((kernel, wtext, line.strip().split("//")[0].strip(), since, sorted(map(str, hit))))
        # a later write of the same register replaces the earlier one as the value a reader sees
        window = [(since + states, wr - w, t) for (since, wr, t) in window if since + states < 2 and (wr - w)]
        if w:
            window.append((0, w, line.strip().split("//")[0].strip()))
    return sites


def scan_cndmask_runs(text, limit=3):
    """Performance lint, not a hazard: on gfx950 a run of v_cndmask_b32 in the encodings that read VCC implicitly (e32 / dpp / sdwa)
    costs ~22.7 cycles per instruction from the third one on, unless a plain vector instruction separates them
    (profiles/r05_valu_rates.txt).  Returns (kernel, run length) for every run of `limit` or more."""
    runs, kernel, run = [], None, 0
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            kernel, run = m.group(1), 0
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//", line)
        if not m:
            continue
        mn = m.group(1)
        if re.match(r"v_cndmask_b32_(e32|dpp|sdwa)$", mn):
            run += 1
            continue
        if mn.startswith("v_"):                                   # a vector instruction ends the run; scalar ones and s_nop do not
            if run >= limit:
                runs.append((kernel, run))
            run = 0
    return runs


def disassemble(lib):
    """llvm-objdump --offloading extracts the gfx950 code objects of a fat binary next to it: work on a private copy"""
    import glob
    import shutil
    import tempfile
    d = tempfile.mkdtemp(prefix="trc_isa_")
    try:
        cp = os.path.join(d, "lib.so")
        shutil.copy(lib, cp)
        subprocess.check_call([OBJDUMP, "--offloading", cp], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        text = ""
        for co in sorted(glob.glob(cp + ".*gfx950*")):
            text += subprocess.check_output([OBJDUMP, "-d", "--mcpu=gfx950", co], text=True)
        if not text:
            raise RuntimeError("no gfx950 code object found in " + lib)
        return text
    finally:
        shutil.rmtree(d, ignore_errors=True)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    lib = args[0] if args else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "turbo-range-coder_amd", "libturborc_hip.so")
    text = disassemble(lib)
    sites = scan(text)
    per = {}
    for k, wt, rt, since, hit in sites:
        per.setdefault(k, []).append((wt, rt, since, hit))
    kernels = len(re.findall(r"^[0-9a-f]+ <.+>:", text, flags=re.M))
    for k in sorted(per):
        print("%s: %d sites" % (k, len(per[k])))
        if "-v" in sys.argv:
            for wt, rt, since, hit in per[k][:12]:
                print("    %-60s -> %-60s (%d wait states, %s)" % (wt, rt, since, ",".join(hit)))
    print("%d kernels scanned, %d VALU-write -> VALU-read sites with fewer than two wait states" % (kernels, len(sites)))
    runs = scan_cndmask_runs(text)
    if runs:
        worst = {}
        for k, n in runs:
            worst[k] = (worst.get(k, (0, 0))[0] + 1, max(worst.get(k, (0, 0))[1], n))
        print("performance lint: runs of >= 3 VCC-implicit v_cndmask_b32 (22.7 cycles each on gfx950): %d runs in %d kernels" % (len(runs), len(worst)))
        if "-v" in sys.argv:
            for k in sorted(worst):
                print("    %s: %d runs, longest %d" % (k, worst[k][0], worst[k][1]))
    return 1 if sites else 0


if __name__ == "__main__":
    sys.exit(main())
