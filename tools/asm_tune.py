"""Post-pass over the compiled gfx950 assembly of deferred.hip: VGPR renumbering against operand bank
conflicts in the light loops.

Measured on MI355X (tools/gen_ubench_bank.py, gen_ubench_bank2.py): a VALU instruction that reads three
VGPR source operands whose register numbers all have the same parity issues in twice the time of any
other (v_fma_f32 d, v20, v22, v24: 1.81 ns per wave-instruction per SIMD; v20, v23, v24: 0.93; the
accumulator of v_fmac counts as a source; a register repeated in two slots counts twice).  The register
allocator does not know this, so about a quarter of the three-source instructions of the light loop
(its FMAs) come out conflicting.

The pass permutes VGPR numbers per kernel: every reference to a register is renamed consistently over
the whole function, so the instruction stream is unchanged except for register numbers and the results
are bit-identical.  Registers that appear in multi-register operands (v[a:b]: loads, stores, LDS
reads) and v0 (the work-item id the hardware delivers) keep their numbers; the others are
re-assigned, preserving how many even and odd numbers are in use, to minimise the number of conflicting
instructions inside the loops that hold the transcendental groups (the light loops).

Result on MI355X: 23 -> 4 conflicting instructions in the loop, and a kernel time within 1 % of the
untouched build - in a mixed instruction stream at four waves per SIMD the extra operand-fetch cycle is
hidden.  The build therefore runs this pass only on request (VKR_ASM_TUNE=1); it is kept as the route
for edits of the compiled loop that the compiler cannot be steered to (schedule, register assignment).
"""
from __future__ import annotations

import random
import re

_VREG = re.compile(r"\bv(\d+)\b")
_VTUPLE = re.compile(r"\bv\[(\d+):(\d+)\]")
_SRC_SLOT = re.compile(r"^-?\|?v(\d+)\|?$")
_TRANS = ("v_rsq_f32", "v_sqrt_f32", "v_rcp_f32")


def _code(line: str) -> str:
    return line.split(";", 1)[0].strip()


def _is_instruction(line: str) -> bool:
    c = _code(line)
    return bool(c) and not c.startswith(".") and not c.endswith(":")


def source_slots(code: str) -> list[int] | None:
    """VGPR numbers in the source operand slots of a plain VALU instruction (None for anything else)."""
    parts = code.split(None, 1)
    if len(parts) < 2:
        return None
    op, rest = parts
    if not op.startswith("v_") or "[" in rest:
        return None
    ops = [o.strip() for o in rest.split(",")]
    # strip trailing modifiers from the last operand ("v24 clamp", "v3 mul:2")
    ops = [o.split()[0] if o else o for o in ops]
    if not ops or not _SRC_SLOT.match(ops[0]):
        return None
    dst, srcs = ops[0], ops[1:]
    slots = [int(m.group(1)) for m in (_SRC_SLOT.match(s) for s in srcs) if m]
    if op.startswith(("v_fmac_", "v_mac_")):
        slots.append(int(_SRC_SLOT.match(dst).group(1)))
    return slots


def conflicts(code: str) -> bool:
    s = source_slots(code)
    return s is not None and len(s) == 3 and len({r & 1 for r in s}) == 1


def _functions(lines: list[str]):
    """(start, end) line ranges of kernel bodies: from the `name:` label to its .Lfunc_end."""
    out, start = [], None
    for i, l in enumerate(lines):
        if re.match(r"^[A-Za-z_][\w$.]*:\s*(;.*)?$", l) and not l.startswith(".L"):
            start = i
        elif l.startswith(".Lfunc_end") and start is not None:
            out.append((start, i))
            start = None
    return out


def _hot_lines(lines: list[str], lo: int, hi: int) -> set[int]:
    """Indices of instruction lines inside innermost loops that contain transcendental groups."""
    labels = {}
    for i in range(lo, hi):
        m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
        if m:
            labels[m.group(1)] = i
    hot: set[int] = set()
    for i in range(lo, hi):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", _code(lines[i]))
        if not m or m.group(1) not in labels or labels[m.group(1)] >= i:
            continue
        body = range(labels[m.group(1)] + 1, i)
        codes = [_code(lines[j]) for j in body]
        if any("s_cbranch" in c or "s_branch" in c for c in codes):
            continue
        if sum(c.startswith(_TRANS) for c in codes) >= 4:
            hot.update(j for j in body if _is_instruction(lines[j]))
    return hot


def _retune_function(lines: list[str], lo: int, hi: int, rng: random.Random) -> dict:
    hot = _hot_lines(lines, lo, hi)
    if not hot:
        return {"hot": 0}
    used, pinned = set(), {0}
    for i in range(lo, hi):
        if not _is_instruction(lines[i]):
            continue
        c = _code(lines[i])
        for a, b in _VTUPLE.findall(c):
            pinned.update(range(int(a), int(b) + 1))
            used.update(range(int(a), int(b) + 1))
        used.update(int(r) for r in _VREG.findall(c))
    free = sorted(used - pinned)
    triples = []
    for i in hot:
        s = source_slots(_code(lines[i]))
        if s is not None and len(s) == 3:
            triples.append(tuple(s))
    parity = {r: r & 1 for r in used}

    def cost(par):
        return sum(1 for t in triples if par[t[0]] == par[t[1]] == par[t[2]])

    before = cost(parity)
    by_reg: dict[int, list[int]] = {}
    for k, t in enumerate(triples):
        for r in set(t):
            by_reg.setdefault(r, []).append(k)

    def local(par, regs):  # cost of the triples touching any of `regs`
        ks = {k for r in regs for k in by_reg.get(r, ())}
        return sum(1 for k in ks if par[triples[k][0]] == par[triples[k][1]] == par[triples[k][2]])

    best, best_cost = dict(parity), before
    for _ in range(20):  # restarts of a pairwise-swap descent that keeps the even / odd counts
        par = dict(best)
        for _ in range(3):  # perturb
            ev = [r for r in free if par[r] == 0]
            od = [r for r in free if par[r] == 1]
            if ev and od:
                a, b = rng.choice(ev), rng.choice(od)
                par[a], par[b] = 1, 0
        improved = True
        while improved:
            improved = False
            ev = [r for r in free if par[r] == 0]
            od = [r for r in free if par[r] == 1]
            rng.shuffle(ev)
            rng.shuffle(od)
            for a in ev:
                for b in od:
                    if par[a] != 0 or par[b] != 1:
                        continue
                    old = local(par, (a, b))
                    par[a], par[b] = 1, 0
                    if local(par, (a, b)) < old:
                        improved = True
                    else:
                        par[a], par[b] = 0, 1
        c = cost(par)
        if c < best_cost:
            best, best_cost = dict(par), c
        if best_cost == 0:
            break
    to_odd = [r for r in free if parity[r] == 0 and best[r] == 1]
    to_even = [r for r in free if parity[r] == 1 and best[r] == 0]
    assert len(to_odd) == len(to_even)
    mapping = {}
    for a, b in zip(to_odd, to_even):  # exchange their numbers
        mapping[a], mapping[b] = b, a
    if mapping:
        def rename(m):
            r = int(m.group(1))
            return f"v{mapping.get(r, r)}"
        for i in range(lo, hi):
            if _is_instruction(lines[i]):
                code, sep, comment = lines[i].partition(";")
                lines[i] = _VREG.sub(rename, code) + sep + comment
    after = sum(1 for i in hot if conflicts(_code(lines[i])))
    return {"hot": len(hot), "three_source": len(triples), "before": before, "after": after, "renamed": len(mapping)}


def retune(asm_text: str, seed: int = 1) -> tuple[str, list[dict]]:
    """Returns the renumbered assembly and one report per kernel that has a light loop."""
    lines = asm_text.split("\n")
    rng = random.Random(seed)
    reports = []
    for lo, hi in _functions(lines):
        opcodes_before = [_code(l).split()[0] for l in lines[lo:hi] if _is_instruction(l)]
        rep = _retune_function(lines, lo, hi, rng)
        opcodes_after = [_code(l).split()[0] for l in lines[lo:hi] if _is_instruction(l)]
        assert opcodes_before == opcodes_after
        if rep.get("hot"):
            rep["kernel"] = lines[lo].split(":")[0]
            reports.append(rep)
    return "\n".join(lines), reports


def compile_with_pass(hipcc: str, src: str, obj: str, flags: list[str], verbose: bool = False) -> list[dict]:
    """hipcc's own steps, split so that the device assembly can be edited: device code to assembly, retune(),
    assemble + link to a code object, bundle it, compile the host side with that bundle embedded.  The result is
    an ordinary HIP object file that can replace lib/deferred.o before the link step of vk-renderer_amd/build.py.
    An experiment only: the renumbered kernel measured within 1 % of the plain build."""
    import os
    import subprocess
    import sys
    llvm_bin = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
    stem = obj[:-2]
    dev_s, tuned_s, dev_o, dev_co, fatbin = stem + ".gfx950.s", stem + ".gfx950.tuned.s", stem + ".gfx950.o", stem + ".gfx950.co", stem + ".hipfb"

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)

    run([hipcc, "--cuda-device-only", "-S", src, "-o", dev_s] + flags)
    text, reports = retune(open(dev_s).read())
    open(tuned_s, "w").write(text)
    run([os.path.join(llvm_bin, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", tuned_s, "-o", dev_o])
    run([os.path.join(llvm_bin, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", dev_co, dev_o])
    run([os.path.join(llvm_bin, "clang-offload-bundler"), "-type=o", "-bundle-align=4096",
         "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null", f"-input={dev_co}", f"-output={fatbin}"])
    run([hipcc, "--cuda-host-only", "-c", src, "-o", obj] + flags + ["-Xclang", "-fcuda-include-gpubinary", "-Xclang", fatbin])
    return reports


if __name__ == "__main__":
    import sys
    text, reps = retune(open(sys.argv[1]).read())
    for r in reps:
        print(r)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
