#!/usr/bin/env python3
"""Build-time guard for the kernels that count their own vector-memory waits.

Several kernels issue loads through inline asm (`global_load_dword* %0` with a "=v" output, LDS-DMA) and wait for them
with hand-counted `s_waitcnt vmcnt(N)`.  The compiler treats such an output register as defined when the asm statement
is issued, so nothing but the code's structure keeps it from copying, re-using or spilling the register before the load
has landed.  This tool reads the gfx950 code objects out of the built libnerf_hip.so and checks, per kernel,

  1. resources: no scratch memory (a spill of a load's destination to memory between the load and its wait would store
     a stale value); where every load is hand-counted (2.) no register spill of any kind.  One kernel is allowed a
     few spill slots (PROLOGUE_SCRATCH: it loads nothing into registers by hand), but only outside its layer loop: no
     scratch instruction may sit between its first and its last MFMA;
  2. the instruction stream: every vector-memory load into registers is tracked in issue order; `s_waitcnt vmcnt(N)`
     retires all but the N youngest vector-memory operations (loads, stores and LDS-DMA count together, in order); an
     instruction that reads or overwrites a register an outstanding load still has to write is reported.  The listing
     is walked linearly, then once more (operations in flight at the end of the first walk meet the loop bodies again),
     which covers the back edges of the software-pipelined loops.

usage: python tools/kernel_guard.py [--lib path] [--kernel substring ...]     exit status 1 on a finding
"""
import argparse
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
# kernels with hand-counted waits (substrings of the demangled-ish symbol)
GUARDED = ("wgrad_kernel", "wgrad16_kernel", "mlp16_bf16_kernel", "mlp_split_kernel", "mlp_split_train_kernel", "mlp_split_dgrad_kernel",
           "mlp_bf16_kernel",
           "mlp_dgrad_bf16_kernel")
# of these, the kernels whose EVERY load into registers is hand-counted inline asm: only there is the linear walk of the
# instruction stream meaningful (compiler-managed loads sit in branches a linear walk cannot follow, and the compiler
# places their waits itself); the LDS-DMA of the other kernels has no destination register to watch
STREAM_CHECKED = ("wgrad_kernel",)
# kernel -> bytes of scratch per lane it may use in front of its first MFMA (loop invariants of the encoding phase)
PROLOGUE_SCRATCH = {"mlp_split_train_kernel": 64}


def code_objects(lib_path):
    """gfx950 ELF images of every offload bundle in the library."""
    blob = open(lib_path, "rb").read()
    out = []
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob):
        p = m.start()
        n = struct.unpack_from("<Q", blob, p + 24)[0]
        o = p + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, o)
            o += 24
            triple = blob[o:o + tl].decode()
            o += tl
            if "gfx950" in triple and size:
                out.append(blob[p + off:p + off + size])
    return out


def kernel_metadata(elf_path):
    """[{name, vgpr_count, agpr_count, private_segment_fixed_size, vgpr_spill_count, sgpr_spill_count}]"""
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", elf_path], capture_output=True, text=True,
                         check=True).stdout
    kernels, cur = [], None
    for line in txt.splitlines():
        m = re.match(r"\s+(-\s+)?\.(\w+):\s+(\S+)\s*$", line)
        if not m:
            continue
        dash, key, val = m.groups()
        if key == "agpr_count" and dash:      # first key of a kernel record (keys are sorted)
            cur = {}
            kernels.append(cur)
        if cur is not None and key in ("name", "vgpr_count", "agpr_count", "private_segment_fixed_size",
                                       "vgpr_spill_count", "sgpr_spill_count"):
            cur[key] = val if key == "name" else int(val)
    return [k for k in kernels if "name" in k]


_REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in _REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def check_stream(lines):
    """lines: disassembly of one kernel.  Returns a list of findings."""
    findings = []
    outstanding = []      # in issue order: (line number, text, destination registers or empty set)
    for sweep in (0, 1):
        for no, text in lines:
            ins = text.split()[0]
            ops = text[len(ins):]
            if ins == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", text)
                if m:
                    keep = int(m.group(1))
                    outstanding = outstanding[len(outstanding) - keep:] if keep else []
                continue
            touched = regs_of(ops)
            for (lno, ltext, dst) in outstanding:
                hit = dst & touched
                if hit:
                    findings.append("line %d `%s` touches %s while the load of line %d `%s` may still be in flight"
                                    % (no, text.strip(), sorted(hit)[:4], lno, ltext.strip()))
            is_vmem = ins.startswith(("global_", "buffer_", "flat_", "scratch_"))
            if is_vmem:
                dst = set()
                if "_load" in ins and "_lds_" not in ins and " lds" not in text:
                    first = ops.split(",")[0]
                    dst = regs_of(first)
                outstanding.append((no, text, dst))
        if not any(d for _, _, d in outstanding):
            break
    # the same instruction is met twice on the second sweep: report each finding once
    return sorted(set(findings))


def scratch_in_layer_loop(lines):
    """Findings for scratch instructions between the first and the last MFMA of a kernel's listing."""
    mfma = [i for i, (_, text) in enumerate(lines) if text.startswith("v_mfma")]
    if not mfma:
        return []
    return ["line %d `%s`: scratch access inside the layer loop" % (no, text.strip())
            for i, (no, text) in enumerate(lines) if mfma[0] < i < mfma[-1] and text.startswith("scratch_")]


def kernel_streams(elf_path):
    txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", elf_path], capture_output=True,
                         text=True, check=True).stdout
    streams, cur = {}, None
    for no, line in enumerate(txt.splitlines()):
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = streams.setdefault(m.group(1), [])
            continue
        if cur is not None and line.startswith(("\t", " ")) and line.strip():
            body = line.split("//")[0].strip()
            if body:
                cur.append((no, body))
    return streams


def run(lib_path, only=()):
    report, bad = [], 0
    with tempfile.TemporaryDirectory() as tmp:
        for i, image in enumerate(code_objects(lib_path)):
            path = os.path.join(tmp, "co%d.elf" % i)
            with open(path, "wb") as f:
                f.write(image)
            meta = {k["name"]: k for k in kernel_metadata(path)}
            streams = None
            for name, k in sorted(meta.items()):
                if not any(g in name for g in GUARDED) or (only and not any(o in name for o in only)):
                    continue
                probs = []
                strict = any(g in name for g in STREAM_CHECKED)      # elsewhere spills stay in registers (scratch 0 B)
                allowed = max([v for g, v in PROLOGUE_SCRATCH.items() if g in name] + [0])
                scratch = k.get("private_segment_fixed_size", 0)
                if scratch and scratch <= allowed:
                    if streams is None:
                        streams = kernel_streams(path)
                    probs += scratch_in_layer_loop(streams.get(name, []))
                elif scratch or (strict and (k.get("vgpr_spill_count", 0) or k.get("sgpr_spill_count", 0))):
                    probs.append("scratch %d B, %d VGPR spills, %d SGPR spills" % (k.get("private_segment_fixed_size", 0),
                                                                                   k.get("vgpr_spill_count", 0),
                                                                                   k.get("sgpr_spill_count", 0)))
                if strict:
                    if streams is None:
                        streams = kernel_streams(path)
                    probs += check_stream(streams.get(name, []))
                report.append((name, k.get("vgpr_count", 0), k.get("agpr_count", 0), probs))
                bad += bool(probs)
    return report, bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "nerf-loc_amd", "lib", "libnerf_hip.so"))
    ap.add_argument("--kernel", action="append", default=[])
    a = ap.parse_args()
    report, bad = run(a.lib, a.kernel)
    for name, v, ag, probs in report:
        print("%-90s vgpr %3d agpr %3d  %s" % (name[:90], v, ag, "ok" if not probs else "FINDINGS"))
        for p in probs[:10]:
            print("    " + p)
    print("%d kernels checked, %d with findings" % (len(report), bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
