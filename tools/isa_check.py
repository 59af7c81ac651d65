#!/usr/bin/env python3
"""tools/isa_check.py [cvr_spmv.s] -- build-time guard of the hand-scheduled register ring of spmv_ilv_kernel and spmv_gang_kernel (cvr_amd/csrc/cvr_spmv.hip).

The kernel keeps its in-flight loads in v[cap..255], registers the compiler does not allocate (amdgpu_num_vgpr(cap)), issues them from
asm statements and waits with COUNTED s_waitcnt vmcnt(K).  That is correct only while, between the run-in and the end of the loop,
  * every vector-memory instruction is one of those asm statements' loads (a compiler-issued one, or a spill to scratch, would shift
    the count and the kernel would consume ring registers that have not landed: silently wrong sums on some inputs), and
  * the compiler itself never touches a ring register.
This script compiles cvr_spmv.hip to gfx950 assembly (or reads the file given) and checks, for EVERY instantiation of the kernel:
  1. private segment (scratch) size 0, no VGPR / SGPR spills (kernel metadata);
  2. between `; CVR_RING_BEGIN cap=N` and `; CVR_RING_END` every buffer_/global_/flat_/scratch_ instruction is a buffer_load_dword*
     whose destination lies in v[N..255], and stands inside an asm statement;
  3. outside asm statements no instruction of the kernel up to `; CVR_RING_END` names a register v[N..255] (behind it the ring is dead).
Exit status 0 = all instantiations pass (they are listed); 1 = a violation (printed with its line).  `make -C cvr_amd/csrc isa-check`
and __graft_entry__.build() run it; tests/test_host_cpu.py::test_ring_kernel_isa_guard runs it and checks that a broken kernel fails.
  HIPCC_EXTRA="-DX=1 ..." adds compiler flags (the test's way of breaking the kernel on purpose)."""
import os
import re
import shlex
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "cvr_amd", "csrc", "cvr_spmv.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only", "-S"]      # (the Makefile's HIPFLAGS)

NUM = r"(0x[0-9a-fA-F]+|\d+)"
VREG = re.compile(r"\bv(\d+)\b|\bv\[" + NUM + r"(?::" + NUM + r")?\]")          # v7, v[4:5], v[0x80:0x83] (asm operands print in hex), v[0x66]
VMEM = re.compile(r"^\s*(buffer_|global_|flat_|scratch_)\w+")


def compile_to_asm(src=SRC):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    extra = shlex.split(os.environ.get("HIPCC_EXTRA", ""))
    r = subprocess.run([HIPCC] + FLAGS + extra + [src, "-o", out], capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-4000:])
        print(f"isa_check: FAIL: {HIPCC} did not compile {src}")
        raise SystemExit(1)
    return out


def kernels(lines):
    """{symbol: (first line, last line)} of the function bodies of spmv_ilv_kernel instantiations"""
    out, cur, start = {}, None, 0
    for i, l in enumerate(lines):
        m = re.match(r"^(_ZN3cvr\S*spmv_(?:ilv|gang)_kernel\S*):", l)
        if m:
            cur, start = m.group(1), i
        elif cur and l.startswith(".Lfunc_end"):
            out[cur] = (start, i)
            cur = None
    return out


def metadata(lines):
    """{symbol: {key: int}} from the amdhsa.kernels notes at the end of the file (one YAML list item per kernel)"""
    out, item = {}, {}
    in_notes = False
    for l in lines:
        if "amdhsa.kernels:" in l:
            in_notes = True
            continue
        if not in_notes:
            continue
        if re.match(r"^\s*-\s", l) and not re.match(r"^\s{4,}-\s", l):      # a new kernel entry ("  - .agpr_count: ...")
            if item.get("name"):
                out[item["name"]] = item
            item = {}
        m = re.match(r"^\s*-?\s*\.(name|private_segment_fixed_size|vgpr_spill_count|sgpr_spill_count|vgpr_count):\s*(\S+)", l)
        if m:
            item[m.group(1)] = m.group(2) if m.group(1) == "name" else int(m.group(2))
    if item.get("name"):
        out[item["name"]] = item
    return out


def highest_vreg(text):
    hi = -1
    for m in VREG.finditer(text):
        hi = max(hi, int(m.group(1)) if m.group(1) is not None else int(m.group(3) or m.group(2), 0))
    return hi


def first_reg(text):
    """first register operand of an instruction = its destination (loads)"""
    m = VREG.search(text)
    if not m:
        return None
    return int(m.group(1)) if m.group(1) is not None else int(m.group(2), 0)


def check(path, expected):
    lines = open(path).read().splitlines()
    ks, md = kernels(lines), metadata(lines)
    errors = []
    # both ring kernels must be there; their number follows the launch code's template dispatch (float / double x dictionary x 16-bit tags x
    # non-temporal stream loads: 16 each today) and is only pinned when the caller asks for it (ISA_CHECK_EXPECTED) -- a 17th instantiation, or a
    # compiler that merges two, is not an error of the ring
    for kname in ("spmv_ilv_kernel", "spmv_gang_kernel"):
        if not any(kname in sym for sym in ks):
            errors.append(f"no instantiation of {kname} in the assembly")
    if expected and len(ks) != expected:
        errors.append(f"{len(ks)} instantiations of the ring kernels in the assembly, {expected} expected")
    for sym, (a, b) in sorted(ks.items()):
        short = re.search(r"(spmv_(?:ilv|gang)_kernel)I(\w+?)EEv", sym)
        name = f"{short.group(1) if short else 'spmv_ilv_kernel'}<{short.group(2) if short else '?'}>"
        m = md.get(sym, {})
        for key in ("private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count"):
            if m.get(key, -1) != 0:
                errors.append(f"{name}: .{key} = {m.get(key, 'missing')} (must be 0: a spill inside the loop shifts the counted vmcnt)")
        # the cap first: the compiler's own instructions are checked over the whole body
        cap = None
        for i in range(a, b):
            mb = re.search(r";\s*CVR_RING_BEGIN cap=(0x[0-9a-fA-F]+|\d+)", lines[i])
            if mb:
                cap = int(mb.group(1), 0)
                break
        if cap is None:
            errors.append(f"{name}: no CVR_RING_BEGIN marker")
            continue
        in_asm, region, closed = False, False, False
        for i in range(a, b):
            l = lines[i]
            if "#ASMSTART" in l:
                in_asm = True
                continue
            if "#ASMEND" in l:
                in_asm = False
                continue
            if "CVR_RING_BEGIN" in l:
                region = True
                continue
            if "CVR_RING_END" in l:
                region, closed = False, True
                continue
            code = l.split(";")[0]
            if not code.strip() or code.strip().startswith(".") or code.rstrip().endswith(":"):
                continue
            if region and VMEM.match(code):
                ok = in_asm and re.match(r"^\s*buffer_load_dword(x[234])?\b", code) and (first_reg(code) or 0) >= cap
                if not ok:
                    errors.append(f"{name}: vector-memory instruction inside the ring region that is not a ring load ({path}:{i + 1}): {code.strip()}")
            # (behind CVR_RING_END the ring is dead and its registers are the compiler's again -- the gang kernel's fused combine uses a few; a value
            # that lived THROUGH the region in one of them would have been named in front of it, which this rule still catches)
            if not in_asm and not closed and highest_vreg(code) >= cap:
                errors.append(f"{name}: the compiler uses a ring register ({path}:{i + 1}): {code.strip()}")
        if not closed or region:
            errors.append(f"{name}: CVR_RING_BEGIN without CVR_RING_END")
        print(f"isa_check: {name}: scratch {m.get('private_segment_fixed_size')}, vgpr spills {m.get('vgpr_spill_count')}, sgpr spills {m.get('sgpr_spill_count')}, "
              f"vgprs {m.get('vgpr_count')}, ring from v{cap}")
    # the streaming passes beside the ring kernels live on occupancy: a change that makes the compiler index a register array dynamically shows up as scratch and a
    # register count that halves the wavefronts per SIMD long before any test fails (round 6: the combine pass's bitmap form went from 46 / 76 registers to 138 / 235
    # + 1 156 bytes of scratch, and from 35 to 55 us, with every parity test green)
    for sym, m in sorted(md.items()):
        if "combine_bits_kernel" not in sym and "combine_kernel" not in sym:
            continue
        short = re.search(r"(combine(?:_bits)?_kernel\w*)", sym)
        name = short.group(1)[:60] if short else sym[:60]
        if m.get("private_segment_fixed_size", -1) != 0 or m.get("vgpr_count", 999) > 128:
            errors.append(f"{name}: scratch {m.get('private_segment_fixed_size')}, vgprs {m.get('vgpr_count')} (a streaming pass: no scratch, at most 128 registers)")
    return errors


def main():
    expected = int(os.environ.get("ISA_CHECK_EXPECTED", "0"))           # 0: any number (at least one of each kernel)
    path = sys.argv[1] if len(sys.argv) > 1 else compile_to_asm()
    errors = check(path, expected)
    for e in errors[:40]:
        print("isa_check: FAIL:", e)
    if len(sys.argv) <= 1:
        os.unlink(path)
    if errors:
        return 1
    print("isa_check: ok")
    return 0


if __name__ == "__main__":
    sys.exit(main())
