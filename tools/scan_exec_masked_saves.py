#!/usr/bin/env python3
"""Static detector of the code-generation defect behind the round-2 'out-of-line device call' failure.

ROCm 7.2's hipcc (gfx950), given a kernel that CALLS a device function, splits the live ranges of the values that must survive the call:
VGPR copies into call-preserved registers before the call, copies back after it.  It places the saving copies at the top of the block
that follows a `for (k = tid; k < N; k += NT)` region — BEFORE the `s_or_b64 exec, exec, sX` that re-enables the lanes — while the copies back run
with all lanes on.  That block is reached with the `tid < N` subset of the lanes or, from the loop's exit, with NO lane (a divergent loop leaves
exec = 0 until the restore): the saves store nothing and every lane gets garbage for per-lane values (1/dt, N as a double, LDS addresses):
NaNs, wrong results, or a memory fault from a wild address.

The scan: in the disassembly of every gfx950 code object of a library, for every `s_*_saveexec_b64 sX` / `s_xor_b64 sX, exec, ..` followed
by `s_cbranch_execz L`, look at block L up to its `s_or_b64 exec, exec, sX`; any instruction in between that writes a VGPR is such a
copy / reload / rematerialisation executed under the narrowed mask.  Validated on the four experimental builds of tools/run_outline_abi.sh:
it flags exactly the kernels that returned garbage on the GPU (variants A, B, D) and nothing in the build that worked (C, the product).

usage: scan_exec_masked_saves.py LIBRARY_OR_DISASSEMBLY...     (exit code 1 if anything is flagged)"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
VDEF = re.compile(r"^(v_(?!cmp|readlane|readfirstlane|writelane|nop)|scratch_load|global_load|ds_read|ds_bpermute|flat_load|buffer_load)")
SAVE = (re.compile(r"^s_(?:and|andn2|or|xor)_saveexec_b64 (s\[\d+:\d+\]|vcc),"), re.compile(r"^s_xor_b64 (s\[\d+:\d+\]), exec, "))


def scan_text(text):
    cur, ins = None, []
    for l in text.splitlines():
        m = re.match(r"^([0-9a-f]{16}) <(\S+)>:", l)
        if m:
            cur = m.group(2)
            continue
        m = re.match(r"^\t(\S.*?)\s+// ([0-9A-F]+):", l)
        if m and cur:
            ins.append((int(m.group(2), 16), m.group(1).strip(), cur, l))
    base = {}
    for a, _, f, _ in ins:
        base.setdefault(f, a)
    idx = {a: i for i, (a, _, _, _) in enumerate(ins)}
    hits, calls = [], []
    for i, (a, t, f, l) in enumerate(ins):
        if re.match(r"^s_(swappc|call)_b64\b", t):
            calls.append((f, hex(a)))
        m = SAVE[0].match(t) or SAVE[1].match(t)
        if not m:
            continue
        sx = m.group(1)
        for j in range(i + 1, min(i + 4, len(ins))):
            tj = ins[j][1]
            if tj.startswith("s_cbranch_execz"):
                mm = re.search(r"<(\S+?)(?:\+0x([0-9a-f]+))?>\s*$", ins[j][3])
                k = idx.get(base[mm.group(1)] + int(mm.group(2) or "0", 16)) if mm and mm.group(1) in base else None
                if k is None:
                    break
                pre = []
                for q in range(k, min(k + 60, len(ins))):
                    tq = ins[q][1]
                    if tq.startswith("s_or_b64 exec, exec, " + sx):
                        bad = [p for p in pre if VDEF.match(p)]
                        if bad:
                            hits.append((f, hex(ins[k][0]), bad))
                        break
                    if re.match(r"^(s_cbranch|s_branch|s_endpgm|s_setpc|s_swappc)", tq) or re.match(r"^s_\w+ exec,", tq) or "saveexec" in tq:
                        break
                    pre.append(tq)
                break
            if re.match(r"^(s_branch|s_cbranch)", tj):
                break
    return hits, calls, len(base)


def code_objects(lib, workdir):
    """gfx950 code objects embedded in a host library / object (llvm-objdump --offloading unbundles next to its input: work on a copy)."""
    dst = os.path.join(workdir, os.path.basename(lib))
    shutil.copy(lib, dst)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", dst], capture_output=True, text=True, check=True)
    return sorted(os.path.join(workdir, f) for f in os.listdir(workdir) if f.startswith(os.path.basename(lib) + ".") and "amdgcn-amd-amdhsa--gfx950" in f)


def scan_library(lib):
    """[(code object, kernels+functions, masked-save hits, device calls)]"""
    out = []
    with tempfile.TemporaryDirectory() as w:
        for co in code_objects(lib, w):
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout
            hits, calls, nfun = scan_text(dis)
            out.append((os.path.basename(co), nfun, hits, calls))
    return out


def main():
    bad = 0
    for p in sys.argv[1:]:
        if p.endswith((".dis", ".txt")):                       # an llvm-objdump -d listing
            hits, calls, nfun = scan_text(open(p).read())
            rows = [(os.path.basename(p), nfun, hits, calls)]
        else:
            rows = scan_library(p)
        for co, nfun, hits, calls in rows:
            print("%s: %d functions, %d device call(s), %d join block(s) with VGPR writes ahead of the exec restore" % (co, nfun, len(calls), len(hits)))
            for f, a, b in hits:
                print("    %s @%s: %s" % (f[:120], a, "; ".join(b[:4])))
            bad += len(hits) + len(calls)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
