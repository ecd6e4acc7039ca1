"""Dev: the "finding 9" audit over hipcc -S output.

`s_waitcnt vmcnt(n)` counts in ISSUE order: it returns when all but the n YOUNGEST vector-memory requests of the wave have
completed.  A load whose result is needed now, issued BEHIND requests that are only needed later (a prefetch parked in
registers, stores), can only be waited for together with everything older -- the software pipeline silently serialises.
The global walk carried exactly that error for three rounds (epilogue parameters loaded behind the next slice's gather).

For every kernel, for every LOOP (a backward branch to a label), the loop body is laid out twice in program order (so that
requests of the previous iteration are in the queue), the vmcnt queue is simulated (loads and stores enqueue; gfx9 has no
separate store counter), and every `s_waitcnt vmcnt(n)` of the second copy is classified:

  forced   = the requests this wait forces to complete (everything but the n youngest)
  consumed = those of `forced` whose destination registers are read before the next vmcnt wait (or the loop's end)
  collateral loads   = forced loads that are NOT consumed in that window, issued BEFORE an older consumed load ... i.e. loads
                       younger than nothing needed: they are waited for only because of the order
  flagged  = a wait with >= 1 consumed load and >= 1 collateral LOAD that is OLDER than the youngest consumed load and is
             itself not consumed until a LATER wait (a parked prefetch in front of a needed load)

Stores in front of a needed load are reported too (`stores_ahead`): the wait covers their write acknowledgements.

    python tools/vmcnt_audit.py /tmp/isa/dense.s [--kernel se_res_mfma] [--all]

A flagged wait whose parked loads are all read again within FAR instructions is batch ORDER noise (a batch of requests in
flight, consumed in a slightly different order than issued: costs at most one request's latency); the ones to read are those
with a load parked for longer that was issued SHORTLY before the wait (it may still be in flight) -- a prefetch for a later
stage sitting in front of a load that is needed now: the finding-9 pattern.

Approximations: program order = text order (forward branches inside a loop body are ignored, the report says when a loop
has them); a destination register overwritten by a later instruction is not tracked; waits with vmcnt(0) at a loop's top
that consume everything are not flagged (nothing parked).
"""
import re
import sys

FAR = 64     # a parked load that is read again only this many instructions behind the wait is a prefetch, not batch order
YOUNG = 150  # ... and it can still be in flight if it was issued fewer instructions than this before the wait (~5 cycles
             # per instruction against 500-900 cycles of L2 / HBM latency)

VM_LOAD = re.compile(r"^\s*(global_load|buffer_load|flat_load|scratch_load)\w*\s+(.*)$")
VM_STORE = re.compile(r"^\s*(global_store|buffer_store|flat_store|scratch_store|global_atomic|buffer_atomic|flat_atomic)\w*\s+(.*)$")
WAIT = re.compile(r"^\s*s_waitcnt\b(.*)$")
LABEL = re.compile(r"^(\.LBB\d+_\d+):")
BRANCH = re.compile(r"^\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)|^\s*s_branch\s+(\.LBB\d+_\d+)")
REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            for i in range(int(m.group(4)), int(m.group(5)) + 1):
                out.add((m.group(3), i))
    return out


def split_operands(ops):
    ops = ops.split(";")[0]
    depth, cur, out = 0, "", []
    for ch in ops:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def parse_kernels(path):
    txt = open(path).read()
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\s*s_endpgm", txt, re.S | re.M):
        yield m.group(1), m.group(2).splitlines()


def demangle(name):
    return re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)[:90]


def source_regs(ln):
    body_txt = ln.split(";")[0]
    mm = re.match(r"^\s*(\S+)\s+(.*)$", body_txt)
    if not mm:
        return set()
    op, ops = mm.group(1), split_operands(mm.group(2))
    srcs = ops if re.match(r"(global_store|buffer_store|flat_store|scratch_store|ds_write|ds_store|s_|global_atomic|v_cmp)", op) else ops[1:]
    if op.startswith(("v_fmac", "v_pk_fmac", "v_mac")):
        srcs = ops
    used = set()
    for o in srcs:
        used |= regs_of(o)
    return used


def is_lds_dma(ops):
    return " lds" in (" " + ops)


def audit_kernel(name, lines, verbose=False):
    labels = {}
    for i, ln in enumerate(lines):
        m = LABEL.match(ln)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, ln in enumerate(lines):
        m = BRANCH.match(ln)
        if m:
            tgt = m.group(1) or m.group(2)
            if tgt in labels and labels[tgt] < i:
                loops.append((labels[tgt], i))
    # innermost-first, drop duplicates with the same head (keep the longest back edge per head)
    by_head = {}
    for h, t in loops:
        by_head[h] = max(by_head.get(h, t), t)
    findings = []
    for h, t in sorted(by_head.items()):
        body = lines[h:t + 1]
        inner_fwd = sum(1 for ln in body if BRANCH.match(ln)) - 1
        seq = body + body
        queue = []      # entries: dict(kind, regs, text, pos, copy)
        events = []     # (pos, forced list, n)
        for pos, ln in enumerate(seq):
            copy = 0 if pos < len(body) else 1
            ml, ms, mw = VM_LOAD.match(ln), VM_STORE.match(ln), WAIT.match(ln)
            if ml:
                ops = split_operands(ml.group(2))
                dst = regs_of(ops[0]) if ops and not is_lds_dma(ml.group(2)) else set()
                queue.append(dict(kind="load", regs=dst, text=ln.strip().split(";")[0][:70], pos=pos, copy=copy))
            elif ms:
                has_ret = "sc0" in ms.group(2) and "atomic" in ms.group(1)
                queue.append(dict(kind="store", regs=set(), text=ln.strip().split(";")[0][:70], pos=pos, copy=copy))
            elif mw:
                mv = re.search(r"vmcnt\((\d+)\)", mw.group(1))
                if not mv:
                    continue
                n = int(mv.group(1))
                forced = queue[:max(0, len(queue) - n)]
                queue = queue[max(0, len(queue) - n):]
                if copy == 1:
                    events.append((pos, forced, n))
        # classify the waits of the second copy
        for ei, (pos, forced, n) in enumerate(events):
            nxt = events[ei + 1][0] if ei + 1 < len(events) else len(seq)
            window = seq[pos + 1:nxt]
            used = set()
            for ln in window:
                used |= source_regs(ln)
            loads = [f for f in forced if f["kind"] == "load"]
            consumed = [f for f in loads if f["regs"] & used]
            if not consumed:
                continue
            youngest_needed = max(f["pos"] for f in consumed)
            parked = [f for f in loads if f not in consumed and f["pos"] < youngest_needed and f["regs"]]
            # how long is a parked load parked?  distance (instructions) from this wait to the first read of its registers;
            # None = not read again before the end of the second copy of the loop body (a prefetch for a later iteration)
            for f in parked:
                f["age"] = pos - f["pos"]   # instructions between the parked load's issue and this wait
                f["use_in"] = None
                for q in range(pos + 1, len(seq)):
                    if f["regs"] & source_regs(seq[q]):
                        f["use_in"] = q - pos
                        break
            stores_ahead = [f for f in forced if f["kind"] == "store" and f["pos"] < youngest_needed]
            if parked or (verbose and stores_ahead):
                findings.append(dict(loop=(h, t), wait_line=h + (pos - len(body)), n=n, consumed=consumed, parked=parked,
                                     stores_ahead=stores_ahead, inner_branches=inner_fwd, body_len=len(body)))
    return findings, len(by_head)


def main(argv):
    want = None
    verbose = "--all" in argv
    paths = []
    it = iter(argv)
    for a in it:
        if a == "--kernel":
            want = next(it)
        elif a == "--all":
            pass
        else:
            paths.append(a)
    for path in paths:
        for name, lines in parse_kernels(path):
            dem = demangle(name)
            if want and want not in dem:
                continue
            findings, nloops = audit_kernel(name, lines, verbose)
            flagged = [f for f in findings if f["parked"]]
            severe = [f for f in flagged if any((c["use_in"] is None or c["use_in"] > FAR) and c["age"] < YOUNG for c in f["parked"])]
            print("%-92s loops %2d  flagged waits %2d  of which a prefetch (not read for > %d instr.) issued < %d instr. before the wait: %2d"
                  % (dem, nloops, len(flagged), FAR, YOUNG, len(severe)))
            for f in findings:
                if not f["parked"] and not verbose:
                    continue
                print("    loop lines %d-%d (%d instr, %d inner branches): s_waitcnt vmcnt(%d) at +%d" % (
                    f["loop"][0], f["loop"][1], f["body_len"], f["inner_branches"], f["n"], f["wait_line"] - f["loop"][0]))
                for c in f["consumed"][:4]:
                    print("        needs   [%s@%d] %s" % ("prev" if c["copy"] == 0 else "this", c["pos"] % f["body_len"], c["text"]))
                for c in f["parked"][:6]:
                    print("        PARKED  [%s@%d] %s   (issued %d instructions before the wait, read again after %s)"
                          % ("prev" if c["copy"] == 0 else "this", c["pos"] % f["body_len"], c["text"], c["age"],
                             "> one iteration" if c["use_in"] is None else c["use_in"]))
                if len(f["parked"]) > 6:
                    print("        ... %d more parked loads" % (len(f["parked"]) - 6))
                if f["stores_ahead"]:
                    print("        (+ %d stores ahead of the needed load)" % len(f["stores_ahead"]))


if __name__ == "__main__":
    main(sys.argv[1:])
