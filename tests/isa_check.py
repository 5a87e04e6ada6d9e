"""TEST INFRASTRUCTURE: static check of compiled gfx950 ISA -- no LDS operation may be in flight at an s_barrier.

gfx950 does not drain the memory counters at `s_barrier`; a workgroup barrier that is meant to separate "everyone has
finished READING this LDS stage" from "someone overwrites it" (LDS-DMA of the next K-step, epilogue strips) only does so if
every wave has waited for its own ds_read results first (`s_waitcnt lgkmcnt(0)`).  The compiler inserts such waits only
in front of the first USE of a result and is free to move a bare `__builtin_amdgcn_s_barrier()` above the MFMAs that use
them -- round 1's run-to-run instability under concurrent hipGraph replays (profiles/r2_notes.md).

`lds_ops_in_flight_at_barriers(asm_text, kernel_regex)` runs a small forward dataflow over each matching function:
state = upper bound of outstanding LGKM-counted LDS instructions (ds_*), joined by max over fall-through and branch
edges; `s_waitcnt lgkmcnt(N)` lowers it to min(state, N) (LDS operations of a wave complete in order).  Returns
{kernel: [(line number, state), ...]} for every s_barrier reached with state > 0.
"""
import re

_LABEL = re.compile(r"^(\.LBB[0-9_]+):")
_BRANCH = re.compile(r"^s_c?branch\w*\s+(\.LBB[0-9_]+)")
_LGKM = re.compile(r"lgkmcnt\((\d+)\)")
CAP = 63


def _functions(asm_text, kernel_regex):
    cur, body = None, []
    for ln, line in enumerate(asm_text.splitlines(), 1):
        l = line.strip()
        m = re.match(r"^(_Z\S+):", l)
        if m and cur is None and re.search(kernel_regex, m.group(1)):
            cur, body = m.group(1), []
            continue
        if cur is not None:
            if l.startswith(".Lfunc_end"):
                yield cur, body
                cur = None
            else:
                body.append((ln, l))


def lds_ops_in_flight_at_barriers(asm_text, kernel_regex):
    out = {}
    for name, body in _functions(asm_text, kernel_regex):
        ins = [(ln, l) for ln, l in body if l and not l.startswith((";", ".amdhsa", ".p2align", ".section", ".type", ".size"))]
        labels = {m.group(1): i for i, (ln, l) in enumerate(ins) for m in [_LABEL.match(l)] if m}
        state_in = [None] * (len(ins) + 1)
        state_in[0] = 0
        work = [0]
        while work:
            i = work.pop()
            st = state_in[i]
            while i < len(ins):
                ln, l = ins[i]
                nxt = st
                stop = False
                if l.startswith("ds_"):
                    nxt = min(CAP, st + 1)
                elif l.startswith("s_waitcnt"):
                    m = _LGKM.search(l)
                    if m:
                        nxt = min(st, int(m.group(1)))
                    elif re.match(r"^s_waitcnt\s+0(x0+)?\s*$", l):
                        nxt = 0
                b = _BRANCH.match(l)
                if b and b.group(1) in labels:
                    t = labels[b.group(1)]
                    if state_in[t] is None or state_in[t] < nxt:
                        state_in[t] = nxt
                        work.append(t)
                    if l.startswith("s_branch"):
                        stop = True
                if l.startswith("s_endpgm"):
                    stop = True
                if stop:
                    break
                i += 1
                if state_in[i] is not None and state_in[i] >= nxt:
                    break
                state_in[i] = nxt
                st = nxt
        bad = [(ln, state_in[i]) for i, (ln, l) in enumerate(ins) if l.startswith("s_barrier") and (state_in[i] or 0) > 0]
        if bad:
            out[name] = bad
    return out


def count_barriers(asm_text, kernel_regex):
    return sum(1 for _, body in _functions(asm_text, kernel_regex) for _, l in body if l.startswith("s_barrier"))
