#!/usr/bin/env python3
"""VGPR pressure of one kernel from its gfx950 assembly (hipcc -S --cuda-device-only): a backward liveness pass over the ISA.

    tools/isa_pressure.py file.s <kernel-symbol-substring> [--top N]

Prints the live-VGPR count at every s_barrier, at the N program points with the highest pressure, and per basic block -- to see WHICH phase of a
persistent kernel the register allocator ran out in (the resource remarks only give the total and the spill count).  Approximations: the first operand
of an instruction that is not a store / compare is a full definition (partial writes under a divergent exec mask or DPP bound control are treated as
kills), scratch (spill) traffic is ignored, AGPRs are not counted."""
import re
import sys

STORE = re.compile(r"^(global_store|scratch_store|ds_write|ds_store|buffer_store|flat_store|global_atomic|ds_add|ds_max|ds_min|s_|v_cmp|v_cmpx|buffer_wbl2|buffer_inv)")
BOTH = re.compile(r"^(v_swap_b32|v_permlane32_swap|v_permlane16_swap)")
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def main():
    path, sym = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 12
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.split(";")[0].strip().endswith(":") and sym in l.split(";")[0] and not l.startswith("."))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start + 1:end + 1]
    # instructions with block structure
    insts = []       # (lineno, mnemonic, defs, uses, text)
    label_at = {}    # label -> index of the next instruction
    for k, raw in enumerate(body):
        l = raw.split(";")[0].strip()
        if not l:
            continue
        if l.endswith(":"):
            label_at[l[:-1]] = len(insts)
            continue
        if l.startswith("."):
            continue
        parts = l.split(None, 1)
        mn = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        if STORE.match(mn):
            d, u = set(), regs(" ".join(ops))
        elif BOTH.match(mn):
            d = u = regs(" ".join(ops))
        else:
            d = regs(ops[0]) if ops else set()
            u = regs(" ".join(ops[1:])) if len(ops) > 1 else set()
            if mn.startswith("v_mac") or mn.startswith("v_fmac") or mn.startswith("v_pk_fmac") or "dpp" in l or mn.startswith("v_dot2c"):
                u = u | d
        insts.append((start + 2 + k, mn, d, u, l))
    n = len(insts)
    succ = [[] for _ in range(n)]
    for i, (_, mn, _, _, l) in enumerate(insts):
        tgt = l.split()[-1] if (mn.startswith("s_cbranch") or mn == "s_branch") else None
        if tgt is not None and tgt in label_at and label_at[tgt] < n:
            succ[i].append(label_at[tgt])
        if mn != "s_branch" and mn != "s_endpgm" and i + 1 < n:
            succ[i].append(i + 1)
    live_in = [set() for _ in range(n)]
    changed = True
    while changed:
        changed = False
        for i in range(n - 1, -1, -1):
            out = set()
            for s in succ[i]:
                out |= live_in[s]
            new = (out - insts[i][2]) | insts[i][3]
            if new != live_in[i]:
                live_in[i] = new
                changed = True
    press = [len(s) for s in live_in]
    print(f"{sym}: {n} instructions, peak {max(press)} live VGPRs")
    print("at the barriers:")
    for i, (ln, mn, _, _, l) in enumerate(insts):
        if mn == "s_barrier":
            print(f"  line {ln}: {press[i]} live")
    if "--at" in sys.argv:      # the live set at a line of the .s file, grouped by the (linearly) closest earlier definition
        at = int(sys.argv[sys.argv.index("--at") + 1])
        i0 = next(i for i in range(n) if insts[i][0] >= at)
        groups = {}
        for r in sorted(live_in[i0]):
            j = next((j for j in range(i0 - 1, -1, -1) if r in insts[j][2]), None)
            key = (insts[j][0], insts[j][4][:70]) if j is not None else (0, "(defined behind this point: loop-carried)")
            groups.setdefault(key, []).append(r)
        print(f"live at line {insts[i0][0]}: {len(live_in[i0])}")
        for (ln, txt), rs in sorted(groups.items()):
            print(f"  line {ln}: {txt}  -> v{rs}")
        return
    print(f"top {top} program points:")
    order = sorted(range(n), key=lambda i: -press[i])
    shown = []
    for i in order:
        if all(abs(i - j) > 40 for j in shown):
            shown.append(i)
            print(f"  line {insts[i][0]}: {press[i]} live   {insts[i][4][:90]}")
        if len(shown) >= top:
            break


if __name__ == "__main__":
    main()
