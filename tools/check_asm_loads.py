#!/usr/bin/env python3
"""Safety check of the TYPED kernels' hand-written vector-memory operations (uwt_kernels.h: load_group_typed): the compiler does
not track loads issued from asm statements, so between such a load and the s_waitcnt that releases it NO instruction may read
or write the load's destination registers.

For every hand-written vector load of a kernel, every path from the load is walked — fall-through AND branch targets, loop
back-edges included — until a `s_waitcnt vmcnt(N)` that covers the load (in-order retirement: a wait for vmcnt <= N releases
a load once at most N younger vector-memory operations have been issued behind it).  On the way
  * no instruction — younger loads, stores and atomics included (their address, data and destination operands) — may mention the
    load's destination registers, and
  * the walk must not reach s_endpgm: a load nobody waits for is a violation.

Input: a compiler listing (`make -C uw-slam_amd/csrc asm`: <unit>.gfx950.s) or — what tests/test_abi_cpu.py checks — the
DISASSEMBLY OF THE SHIPPED LIBRARY (`--shipped <libuwt_hip.so>`: the gfx950 code objects are taken out of the shared object with
llvm-objdump --offloading and disassembled; hipcc's register allocation differs from build to build, so only the binary that
ships can vouch for itself).
usage: check_asm_loads.py <file.s | --shipped lib.so> <mangled-name-substring | --all-typed>"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
TYPED_RE = r"_ZN3uwt10k_residualI\w+?ELi[23]ELb[01]EEEvNS_12ResidualArgsE"   # STREAM & kLoadsTyped, then RAGGED


def shipped_listing(lib):
    """Disassembly of every gfx950 code object inside `lib`, as text."""
    tmp = tempfile.mkdtemp(prefix="uwt_co_")
    try:
        dst = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, dst)
        subprocess.run([OBJDUMP, "--offloading", dst], check=True, capture_output=True, cwd=tmp)
        text = []
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" in f:
                text.append(subprocess.run([OBJDUMP, "-d", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout)
        return "\n".join(text)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def functions(text):
    """{name: [(address-or-index, instruction text, branch target or None)]} for either input format."""
    out = {}
    if re.search(r"^[0-9a-f]{16} <_ZN3uwt", text, re.M):   # llvm-objdump -d
        cur = None
        for l in text.splitlines():
            m = re.match(r"^[0-9a-f]{16} <(\S+)>:", l)
            if m:
                cur = out.setdefault(m.group(1), [])
                continue
            m = re.match(r"^\t(\S.*?)\s*//\s*([0-9A-F]+):\s*[0-9A-F ]+(?:<(\S+?)\+0x([0-9a-f]+)>)?", l)
            if m and cur is not None:
                cur.append((int(m.group(2), 16), m.group(1).strip(), int(m.group(4), 16) if m.group(4) else None))
        for name, ins in out.items():   # branch targets: offsets from the function's first instruction -> addresses
            if ins:
                base = ins[0][0]
                out[name] = [(a, t, base + tgt if tgt is not None and re.match(r"s_c?branch", t) else None) for a, t, tgt in ins]
        return out
    lines = text.splitlines()   # compiler listing
    i = 0
    while i < len(lines):
        m = re.match(r"^(_ZN3uwt\w+):", lines[i])
        if not m:
            i += 1
            continue
        name, body, labels = m.group(1), [], {}
        i += 1
        while i < len(lines) and not lines[i].strip().startswith("s_endpgm"):
            l = lines[i].split(";")[0].rstrip()
            lm = re.match(r"^(\.LBB\d+_\d+):", l)
            if lm:
                labels[lm.group(1)] = len(body)
            elif l.startswith("\t") and l.strip() and not l.strip().startswith("."):
                body.append(l.strip())
            i += 1
        body.append("s_endpgm")
        ins = []
        for k, t in enumerate(body):
            bm = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", t)
            ins.append((k, t, labels.get(bm.group(1)) if bm else None))
        out[name] = ins
    return out


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def mentioned(text):
    """Vector registers an instruction reads or writes.  A packed-f32 instruction names its sources as register PAIRS, but op_sel /
    op_sel_hi decide which register of a pair each lane actually takes (op_sel_hi:[0,1]: both lanes of source 0 read the LOW
    register — a broadcast; the pair's high register is not touched)."""
    m = re.match(r"(v_pk_\w+_f32)\s+(.*)", text)
    if m:
        body = m.group(2)
        sel = re.search(r"op_sel:\[([\d,]+)\]", body)
        sel_hi = re.search(r"op_sel_hi:\[([\d,]+)\]", body)
        ops = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", re.sub(r"\s+(op_sel|op_sel_hi|neg_lo|neg_hi|clamp)\b.*", "", body))]
        lo = [int(x) for x in sel.group(1).split(",")] if sel else [0, 0, 0]
        hi = [int(x) for x in sel_hi.group(1).split(",")] if sel_hi else [1, 1, 1]
        out = regs(ops[0])   # the destination pair is written whole
        for k, o in enumerate(ops[1:]):
            pm = re.fullmatch(r"v\[(\d+):(\d+)\]", o)
            if pm:
                a = int(pm.group(1))
                out |= {a + (lo[k] if k < len(lo) else 0), a + (hi[k] if k < len(hi) else 1)}
            else:
                out |= regs(o)
        return out
    out = set()
    for tok in re.findall(r"v\[\d+:\d+\]|v\d+", text):
        out |= regs(tok)
    return out


VMEM = r"(tbuffer_load|tbuffer_store|global_load|buffer_load|global_store|buffer_store|global_atomic|buffer_atomic|flat_load|flat_store|flat_atomic|scratch_load|scratch_store)"


def check(name, ins):
    pos = {a: k for k, (a, _, _) in enumerate(ins)}
    bad, loads = 0, 0
    for k, (addr, text, _) in enumerate(ins):
        # the hand-written ones: the typed plane loads, the byte gathers, and the intensities' dword load in front of a typed load
        mine = re.match(r"(tbuffer_load|global_load_ubyte|buffer_load_ubyte)", text) or (
            re.match(r"global_load_dword v\d+, v\d+, s\[", text) and any(t.startswith("tbuffer_load") for _, t, _ in ins[k + 1:k + 4]))
        if not mine:
            continue
        dst = regs(text.split()[1].rstrip(","))
        loads += 1
        # every path from the load: (instruction index, younger vector-memory operations issued so far)
        todo, seen = [(k + 1, 0)], set()
        while todo:
            j, younger = todo.pop()
            while True:
                if (j, younger) in seen or j >= len(ins):
                    break
                seen.add((j, younger))
                a, t, target = ins[j]
                m = re.match(r"s_waitcnt .*vmcnt\((\d+)\)", t)
                if m and int(m.group(1)) <= younger:
                    break   # released on this path
                if t.startswith("s_endpgm"):
                    print("%s: the load at %s `%s` reaches s_endpgm without a wait that covers it" % (name[:60], addr, text[:60]))
                    bad += 1
                    break
                if mentioned(t) & dst:
                    print("%s: %s `%s` touches %s of the load at %s `%s` before it is released" % (name[:60], a, t, sorted(mentioned(t) & dst), addr, text[:60]))
                    bad += 1
                    break
                if re.match(VMEM, t):
                    younger = min(younger + 1, 64)
                if target is not None:
                    tj = pos.get(target) if target in pos else (target if isinstance(target, int) and target < len(ins) and ins[target][0] == target else None)
                    if tj is not None:
                        todo.append((tj, younger))
                    if t.startswith("s_branch"):
                        break   # unconditional: no fall-through
                j += 1
    return loads, bad


def main():
    args = sys.argv[1:]
    if args[0] == "--shipped":
        text, key = shipped_listing(args[1]), args[2]
    else:
        text, key = open(args[0]).read(), args[1]
    fns = functions(text)
    names = sorted(n for n in fns if (re.fullmatch(TYPED_RE, n) if key == "--all-typed" else key in n))
    if not names:
        print("no kernel matches %s" % key)
        sys.exit(2)
    tot_loads = tot_bad = 0
    for n in names:
        loads, bad = check(n, fns[n])
        tot_loads += loads
        tot_bad += bad
    print("%d vector loads checked in %d kernels, %d violations" % (tot_loads, len(names), tot_bad))
    sys.exit(1 if tot_bad else 0)


if __name__ == "__main__":
    main()
