"""Build-time check of the wait / tie idiom of the streaming kernels (round 6; profiles/r06_wait_tie_hazard.txt).

The kernels request operand fragments and residual pieces with inline-asm loads (ds_read_b128, global_load_dwordx4) and make them ordinary values at a later inline-asm
`s_waitcnt`.  hipcc does not know that the destination registers are in flight in between: a register copy (v_mov, v_accvgpr_write), a spill or any other instruction it
places between the load and the wait that READS such a register reads stale data -- conv_slice64_head did exactly that once the copies hipcc makes for a "+v" tie were
scheduled in front of the wait.  This script compiles a source to gfx950 assembly and walks every kernel linearly:
  * an inline-asm load (between ;;#ASMSTART / ;;#ASMEND) puts its destination registers in flight (DS and VMEM queues kept apart, in issue order);
  * an s_waitcnt (inline or hipcc's own) retires all but the newest N entries of the queue it names (DS operations retire in order; VMEM: vmcnt(0) retires everything; a
    COUNTED vmcnt moves the older VMEM loads to an "assumed" list -- VGPR loads, LDS-DMA loads and stores do not retire in one order, so a counted wait is an assumption
    the stress tests pin, tests/test_gpu_forward.py::test_streaming_kernels_long_streams_repeat_bit_for_bit, not a guarantee);
  * the walk is linear in the TEXT: behind an unconditional branch the in-flight state is reset (hazards across such an edge are missed, not invented);
  * any instruction OUTSIDE inline asm that reads a register still in flight is reported as a HAZARD (exit code 1); reads behind a counted vmcnt are listed as ASSUMED
    (the residual variants of conv_roll / conv_roll_t / conv_roll_t32, whose deep prefetch rules a drain out).
usage: python tools/isa_wait_lint.py dffinthewild_amd/csrc/dffw_conv_slice.hip [more sources]      (exit code 1 when something is reported; extra compiler
flags through LINT_DEFS, e.g. LINT_DEFS=-DDFFW_SLICE_HAZARD=2 reproduces the round-6 finding: the four copies in front of the wait in each conv_slice64_head instantiation)"""
import os
import re
import subprocess
import sys

REG = re.compile(r"v\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def lint(src):
    asm = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-DDFFW_TILE_PREC=0"] + os.environ.get("LINT_DEFS", "").split() + ["-S", src, "-o", "-"],
                         capture_output=True, text=True, check=True).stdout.split("\n")
    findings = []
    kernel, in_asm = None, False
    ds, vm, vma = [], [], []   # in-flight destination register sets, oldest first; vma: VMEM loads behind a counted vmcnt
    for ln, line in enumerate(asm, 1):
        s = line.strip()
        m = re.match(r"(_Z\w+):", s)
        if m:
            kernel, ds, vm, vma = m.group(1), [], [], []
            continue
        if s.startswith(".Lfunc_end"):
            kernel = None
            continue
        if kernel is None or not s or s.startswith(";") and "ASM" not in s:
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if s.startswith(".") or s.startswith(";"):
            continue
        op = s.split()[0]
        if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            # the textual successor of an unconditional branch is not its successor in execution: what is in flight there is unknown (taken as nothing --
            # a hazard across such an edge is missed rather than invented)
            ds, vm, vma = [], [], []
            continue
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", s)
            if m:
                n = int(m.group(1))
                ds = ds[len(ds) - n:] if n else []
            m = re.search(r"vmcnt\((\d+)\)", s)
            if m and int(m.group(1)) == 0:
                vm, vma = [], []
            elif m:
                vma, vm = vma + vm, []
            continue
        operands = s[len(op):]
        parts = operands.split(",")
        if in_asm:
            if op.startswith("ds_read"):
                ds.append(regs(parts[0]))
            elif (op.startswith("global_load") or op.startswith("buffer_load")) and " lds" not in s:
                vm.append(regs(parts[0]))
            continue
        # an ordinary instruction: its sources are every register operand but the first (stores, compares, v_mfma's accumulator input etc. read all of them)
        reads_all = op.startswith(("global_store", "buffer_store", "ds_write", "scratch_store", "v_cmp", "s_", "global_atomic")) or "_swap" in op
        src_regs = regs(operands if reads_all else ",".join(parts[1:]))
        flying = set().union(*ds, *vm) if (ds or vm) else set()
        hit = src_regs & flying
        if hit:
            findings.append((kernel, ln, s, sorted(hit), "HAZARD"))
        elif vma and src_regs & set().union(*vma):
            findings.append((kernel, ln, s, sorted(src_regs & set().union(*vma)), "ASSUMED"))
            vma = [d - src_regs for d in vma]   # (one report per register)
        # a register the instruction overwrites is no longer "the load's": drop it (the load would clobber the new value -- another bug, not this script's)
        dst = regs(parts[0]) if not reads_all else set()
        ds = [d - dst for d in ds]
        vm = [d - dst for d in vm]
        vma = [d - dst for d in vma]
    return findings


if __name__ == "__main__":
    bad = 0
    for src in sys.argv[1:]:
        f = lint(src)
        haz = [x for x in f if x[4] == "HAZARD"]
        assumed = {}
        for x in f:
            if x[4] == "ASSUMED":
                assumed[x[0]] = assumed.get(x[0], 0) + 1
        print(f"{src}: {len(haz)} read(s) of registers with an asm-issued load in flight; {len(f) - len(haz)} behind a counted vmcnt in {len(assumed)} kernel(s)")
        for kernel, ln, s, hit, kind in haz[:40]:
            name = subprocess.run(["c++filt", kernel], capture_output=True, text=True).stdout.strip()
            print(f"  HAZARD  {name[:70]}  asm line {ln}: {s}   <- v{hit}")
        for kernel, cnt in assumed.items():
            name = subprocess.run(["c++filt", kernel], capture_output=True, text=True).stdout.strip()
            print(f"  assumed {name[:90]}: {cnt} register read(s) behind a counted vmcnt")
        bad += len(haz)
    sys.exit(1 if bad else 0)
