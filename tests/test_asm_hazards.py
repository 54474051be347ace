"""Structural check of the hand-managed hazards in the inline assembly (CPU test; no GPU needed).

LLVM's hazard recognizer does not look inside `asm volatile`, so the wait states gfx940+ requires between a
VALU write of an SGPR and its use are kept true by hand in csrc/bh_force.hip (fast_traverse_asm) and
csrc/bh_tree.hip (mask_levels).  This test disassembles the gfx950 code objects of those translation units
(llvm-objdump, in the image with hipcc), rebuilds the control-flow graph of every kernel and asserts, along
EVERY path into each reader, the distances LLVM's GCNHazardRecognizer enforces for gfx940 (hasVDecCoExecHazard):

  * VALU writes an SGPR -> v_readlane / v_writelane using it as the LANE SELECT: 4 wait states
  * VALU writes an SGPR / VCC -> a VALU instruction reading it as a data operand:  2 wait states
  * VALU writes a VGPR  -> v_readlane / v_readfirstlane reading that VGPR:          1 wait state
  * SALU writes M0      -> v_readlane / v_writelane selecting the lane through M0:  1 wait state (this repo's rule)

A wait state = one issued instruction between writer and reader (s_nop N counts N + 1).  Compiler-generated code
is checked with the same rules (it passes by construction), so the model is exercised on thousands of
instructions, not only on the few hand-written ones.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "nbody-barnes-hut-cuda_amd")
LLVM = "/opt/rocm/lib/llvm/bin"


def _disassemble(tu, tmp):
    obj = os.path.join(PKG, "build", tu + ".o")
    if not os.path.exists(obj):
        subprocess.check_call(["make", "-C", PKG, "-j8", "all"], stdout=subprocess.DEVNULL)
    fat, co = os.path.join(tmp, tu + ".fatbin"), os.path.join(tmp, tu + ".co")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    text = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", co], text=True)
    # the symbol table: labels that share an address appear only once in the disassembly listing
    syms = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-t", co], text=True)
    table = {}
    for ln in syms.splitlines():
        f = ln.split()
        if len(f) >= 5 and re.fullmatch(r"[0-9a-f]{16}", f[0]) and ".text" in f:
            table[f[-1]] = int(f[0], 16)
    return text, table


_REG = re.compile(r"(?<![\w.])(s\[(\d+):(\d+)\]|s(\d+)\b|v\[(\d+):(\d+)\]|v(\d+)\b|vcc_lo\b|vcc_hi\b|vcc\b|m0\b|exec_lo\b|exec_hi\b|exec\b)")


def _regs(operand):
    """set of register names ('s12', 'v3', 'vcc_lo', 'm0', ...) an operand string mentions"""
    out = set()
    for m in _REG.finditer(operand):
        if m.group(2) is not None:
            out |= {f"s{i}" for i in range(int(m.group(2)), int(m.group(3)) + 1)}
        elif m.group(4) is not None:
            out.add(f"s{m.group(4)}")
        elif m.group(5) is not None:
            out |= {f"v{i}" for i in range(int(m.group(5)), int(m.group(6)) + 1)}
        elif m.group(7) is not None:
            out.add(f"v{m.group(7)}")
        elif m.group(1) == "vcc":
            out |= {"vcc_lo", "vcc_hi"}
        elif m.group(1) == "exec":
            out |= {"exec_lo", "exec_hi"}
        else:
            out.add(m.group(1))
    return out


class Ins:
    __slots__ = ("addr", "mn", "ops", "target", "kernel")


def _parse(text):
    """-> list of Ins (program order), {address: index}"""
    ins, at = [], {}
    kernel = None
    for line in text.splitlines():
        m = re.match(r"^([0-9a-f]{16}) <([^>]+)>:", line)
        if m:
            if not m.group(2).startswith("L_"):
                kernel = m.group(2)
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if not m:
            continue
        i = Ins()
        i.mn, ops, i.addr, i.kernel = m.group(1), m.group(2), int(m.group(3), 16), kernel
        i.ops = [o.strip() for o in re.split(r",(?![^\[]*\])", ops)] if ops else []
        i.target = None
        if i.mn.startswith("s_cbranch") or i.mn == "s_branch":
            t = re.search(r"<([^>+]+)\+0x([0-9a-fA-F]+)>\s*$", line)   # numeric form: "<symbol+0xoff>" in the comment
            i.target = (t.group(1), int(t.group(2), 16)) if t else (i.ops[0], 0)
        at[i.addr] = len(ins)
        ins.append(i)
    return ins, at


def _cfg(text, symbols=None):
    ins, at = _parse(text)
    labels, kernel_base = dict(symbols or {}), {}
    for line in text.splitlines():
        m = re.match(r"^([0-9a-f]{16}) <([^>]+)>:", line)
        if m:
            labels[m.group(2)] = int(m.group(1), 16)
            if not m.group(2).startswith("L_"):
                kernel_base[m.group(2)] = int(m.group(1), 16)
    preds = [[] for _ in ins]
    for k, i in enumerate(ins):
        falls = i.mn not in ("s_branch", "s_endpgm", "s_setpc_b64", "s_swappc_b64")
        if falls and k + 1 < len(ins) and ins[k + 1].kernel == i.kernel:
            preds[k + 1].append(k)
        if i.target:
            addr = labels[i.target[0]] + i.target[1]
            preds[at[addr]].append(k)
    return ins, preds


def _is_valu(i):
    return i.mn.startswith("v_")


def _sgpr_dests(i):
    """SGPR / VCC registers a VALU instruction writes"""
    mn = i.mn
    if mn.startswith("v_cmp") or mn.startswith("v_readlane") or mn.startswith("v_readfirstlane"):
        return {r for r in _regs(i.ops[0]) if not r.startswith("v") or r.startswith("vcc")} if i.ops else set()
    if "_co_" in mn or mn.startswith(("v_div_scale", "v_mad_u64_u32", "v_mad_i64_i32")):
        return {r for r in _regs(i.ops[1]) if not (r.startswith("v") and not r.startswith("vcc"))} if len(i.ops) > 1 else set()
    return set()


def _vgpr_dests(i):
    if not _is_valu(i) or not i.ops or i.mn.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
        return set()
    return {r for r in _regs(i.ops[0]) if re.fullmatch(r"v\d+", r)}


def _wait_states(i):
    return (int(i.ops[0], 0) + 1) if i.mn == "s_nop" else 1


def _violations(ins, preds):
    """every (reader index, writer index, rule, distance) that breaks a rule"""
    bad = []
    for k, i in enumerate(ins):
        if not _is_valu(i):
            continue
        lane = i.mn.startswith(("v_readlane", "v_writelane"))
        checks = []  # (registers, writer predicate, required wait states, rule)
        srcs = i.ops[1:] if not i.mn.startswith("v_cmpx") else i.ops
        if lane and len(i.ops) == 3:
            sel = _regs(i.ops[2])
            checks.append(({r for r in sel if re.fullmatch(r"s\d+", r)}, "valu_sgpr", 4, "VALU SGPR write -> lane select"))
            if "m0" in sel:
                checks.append(({"m0"}, "salu_m0", 1, "SALU M0 write -> lane select"))
            data = i.ops[1]
        data_regs = set()
        for o in srcs:
            data_regs |= {r for r in _regs(o) if re.fullmatch(r"s\d+|vcc_lo|vcc_hi", r)}
        if lane and len(i.ops) == 3:
            data_regs -= _regs(i.ops[2]) - _regs(i.ops[1])
        if "_co_" in i.mn or i.mn.startswith(("v_div_scale", "v_mad_u64_u32", "v_mad_i64_i32")):
            data_regs = set()
            for o in i.ops[2:]:
                data_regs |= {r for r in _regs(o) if re.fullmatch(r"s\d+|vcc_lo|vcc_hi", r)}
        if i.mn in ("v_cndmask_b32_e32", "v_addc_co_u32_e32", "v_subb_co_u32_e32", "v_subbrev_co_u32_e32") and \
                not any("vcc" in o for o in i.ops):
            data_regs |= {"vcc_lo", "vcc_hi"}  # implicit VCC read of the e32 encodings
        checks.append((data_regs, "valu_sgpr", 2, "VALU SGPR write -> VALU data read"))
        if i.mn.startswith(("v_readlane", "v_readfirstlane")) and len(i.ops) >= 2:
            checks.append(({r for r in _regs(i.ops[1]) if re.fullmatch(r"v\d+", r)}, "valu_vgpr", 1,
                           "VALU VGPR write -> readlane of it"))
        for regs, kind, need, rule in checks:
            if not regs:
                continue
            # backward search along every path, up to `need` wait states
            stack = [(p, 0) for p in preds[k]]
            seen = {}
            while stack:
                j, dist = stack.pop()
                if dist >= need or seen.get(j, 1 << 30) <= dist:
                    continue
                seen[j] = dist
                w = ins[j]
                if kind == "valu_sgpr" and _is_valu(w) and (_sgpr_dests(w) & regs):
                    bad.append((k, j, rule, dist))
                    continue
                if kind == "valu_vgpr" and (_vgpr_dests(w) & regs):
                    bad.append((k, j, rule, dist))
                    continue
                if kind == "salu_m0" and w.mn.startswith("s_") and w.ops and "m0" in _regs(w.ops[0]) and \
                        not w.mn.startswith(("s_cmp", "s_bitcmp", "s_cbranch", "s_load", "s_waitcnt")):
                    bad.append((k, j, rule, dist))
                    continue
                nd = dist + _wait_states(w)
                for p in preds[j]:
                    stack.append((p, nd))
    return bad


@pytest.fixture(scope="module")
def tmpdir_mod(tmp_path_factory):
    return str(tmp_path_factory.mktemp("dis"))


@pytest.mark.parametrize("tu,must_contain", [("bh_force", ["L_popb_", "L_push3_", "v_writelane_b32", "v_readlane_b32"]),
                                             ("bh_tree", ["v_writelane_b32", "v_cmp"])])
def test_hand_written_assembly_keeps_the_gfx940_wait_states(tu, must_contain, tmpdir_mod):
    text, symbols = _disassemble(tu, tmpdir_mod)
    for s in must_contain:   # the hand-written regions are in the object we are looking at
        assert s in text, s
    ins, preds = _cfg(text, symbols)
    assert len(ins) > 2000
    bad = _violations(ins, preds)
    msg = "\n".join(f"{ins[k].kernel[:60]} @{ins[k].addr:x}: {ins[k].mn} {', '.join(ins[k].ops)}  <- @{ins[j].addr:x} "
                    f"{ins[j].mn} {', '.join(ins[j].ops)}  [{rule}: {d} wait state(s)]" for k, j, rule, d in bad[:20])
    assert not bad, f"{len(bad)} hazard(s):\n{msg}"


def test_the_checker_sees_a_planted_hazard():
    """the model itself: a v_cmp into an SGPR pair followed directly by a v_writelane selecting with it, a VALU
    data read one instruction after the write, and a lane select through M0 right after s_mov m0"""
    text = """
0000000000001000 <k>:
	v_cmp_lt_f32_e64 s[4:5], v0, v1    // 000000001000: 00000000
	v_writelane_b32 v2, s9, s4         // 000000001008: 00000000
	v_readlane_b32 s6, v3, s7          // 000000001010: 00000000
	v_mov_b32_e32 v4, s6               // 000000001018: 00000000
	s_mov_b32 m0, s8                   // 00000000101C: 00000000
	v_writelane_b32 v5, s9, m0         // 000000001020: 00000000
	s_nop 3                            // 000000001028: 00000000
	v_writelane_b32 v2, s9, s4         // 00000000102C: 00000000
	s_endpgm                           // 000000001034: 00000000
"""
    ins, preds = _cfg(text)
    bad = _violations(ins, preds)
    rules = sorted((ins[k].addr, rule) for k, _, rule, _ in bad)
    assert (0x1008, "VALU SGPR write -> lane select") in rules
    assert (0x1018, "VALU SGPR write -> VALU data read") in rules
    assert (0x1020, "SALU M0 write -> lane select") in rules
    assert not any(a == 0x102C for a, _ in rules)   # 6 instructions and an s_nop 3 later: fine
