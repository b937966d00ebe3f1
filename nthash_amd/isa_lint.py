"""ISA lint for the "hidden load" construct of the nthash_amd kernels (gfx950 assembly, as `hipcc -S --cuda-device-only` prints it).

Several kernels issue the next tile's `global_load_*` in inline asm and consume the data after this tile's stores behind a
counted `s_waitcnt vmcnt(n)` that is inline asm as well (nthash_amd/csrc/kmer_kernels.hpp explains why).  hipcc believes the
destination VGPRs of such a load hold their data from the asm statement on, so nothing but the source's discipline keeps it
from copying, spilling or reading them before the data has landed.  A spill is caught by build.py's resource check; this
lint catches the rest, on the ISA the compiler really emitted:

  R1  on every control-flow path from a hidden load to the marker that consumes it (or to a `vmcnt(0)` wait), no
      instruction reads or writes the load's destination registers (a path may also end the program without them);
  R2  every register a `; NTLINT_CONSUME` marker names is the destination of a hidden load of that kernel -- the marker
      prints the registers hipcc holds the values in AT THE WAIT, so a copy made in between shows up as a different name --
      and every hidden load is consumed by some marker;
  R3  walking backwards from a marker, the first vector-memory instruction or vmcnt wait on every path is an inline-asm
      `s_waitcnt vmcnt(n)` (or any `vmcnt(0)`; `; NTLINT_VMCNT_WRAPPED` stands for "64 younger operations were issued").

What it cannot prove is that the n of a counted wait is right (that many younger stores were surely issued): that is
argued at each site and covered by the GPU parity tests.

    python -m nthash_amd.isa_lint file.s [...]        # exit status 1 on a violation
"""
import json
import re
import sys

_LABEL = re.compile(r"^([.\w$]+):")
_REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
_BRANCH = re.compile(r"^s_(?:c?branch\w*)\s+([.\w$]+)")
_VMEM = re.compile(r"^(global_|buffer_|flat_|scratch_|image_)")
_VMCNT = re.compile(r"^s_waitcnt\b.*\bvmcnt\((\d+)\)")
_VMCNT_ALL = re.compile(r"^s_waitcnt\s+(\d+|0x[0-9a-fA-F]+)\s*$")


def _regs(text):
    out = set()
    for m in _REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


class Insn:
    __slots__ = ("text", "line", "in_asm", "regs", "is_marker", "marker_regs", "hidden_dst", "wrapped")

    def __init__(self, text, line, in_asm):
        self.text, self.line, self.in_asm = text, line, in_asm
        self.is_marker = self.wrapped = False
        self.marker_regs = set()
        self.hidden_dst = None
        code, _, comment = text.partition(";")
        code = code.strip()
        if in_asm and "NTLINT_CONSUME" in comment:
            self.is_marker = True
            self.marker_regs = _regs(comment.split("NTLINT_CONSUME", 1)[1])
        if in_asm and "NTLINT_VMCNT_WRAPPED" in comment:
            self.wrapped = True
        self.regs = _regs(code)
        if in_asm and code.startswith("global_load_"):
            first = code.split(None, 1)[1].split(",")[0]
            self.hidden_dst = _regs(first)
        self.text = code if code else text.strip()


def parse_functions(asm_text):
    """{kernel name: [blocks]}; a block = {"label", "insns", "succ"} in text order, fall-through edges included."""
    funcs = {}
    cur = None
    in_asm = False
    name = None
    for ln, raw in enumerate(asm_text.splitlines(), 1):
        line = raw.strip()
        if not line:
            continue
        m = _LABEL.match(line)
        if m and not line.startswith(".L") and not line.startswith("."):
            # a function symbol (kernels are the .amdhsa ones; device functions are all inlined here)
            name = m.group(1)
            cur = [{"label": name, "insns": [], "succ": []}]
            funcs[name] = cur
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end") or line.startswith(".section") or line.startswith(".amdhsa_kernel"):
            if line.startswith(".Lfunc_end"):
                cur = None
            continue
        if m:  # local label: a new basic block
            cur.append({"label": m.group(1), "insns": [], "succ": []})
            continue
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if line.startswith("."):
            continue
        if line.startswith(";"):
            if in_asm and "NTLINT_" in line:
                cur[-1]["insns"].append(Insn(line, ln, True))
            continue
        cur[-1]["insns"].append(Insn(line, ln, in_asm))
    # split blocks at branches and wire the edges
    out = {}
    for fname, blocks in funcs.items():
        split = []
        for b in blocks:
            acc = {"label": b["label"], "insns": [], "succ": []}
            for ins in b["insns"]:
                acc["insns"].append(ins)
                if _BRANCH.match(ins.text) or ins.text.startswith("s_endpgm") or ins.text.startswith("s_setpc") or \
                        ins.text.startswith("s_trap"):
                    split.append(acc)
                    acc = {"label": None, "insns": [], "succ": []}
            split.append(acc)
        index = {b["label"]: i for i, b in enumerate(split) if b["label"]}
        for i, b in enumerate(split):
            last = b["insns"][-1].text if b["insns"] else ""
            m = _BRANCH.match(last)
            if m:
                tgt = index.get(m.group(1))
                if tgt is None:
                    raise ValueError(f"{fname}: branch to unknown label {m.group(1)}")
                b["succ"].append(tgt)
                if not last.startswith("s_branch") and i + 1 < len(split):
                    b["succ"].append(i + 1)
            elif last.startswith("s_endpgm") or last.startswith("s_trap"):
                pass
            elif last.startswith("s_setpc"):
                b["succ"] = [j for j, q in enumerate(split) if q["label"]]  # indirect: anywhere labelled (conservative)
            elif i + 1 < len(split):
                b["succ"].append(i + 1)
        out[fname] = split
    return out


def lint_function(name, blocks):
    """-> (violations, stats)"""
    viol = []
    hidden = [(bi, ii, ins) for bi, b in enumerate(blocks) for ii, ins in enumerate(b["insns"]) if ins.hidden_dst]
    markers = [(bi, ii, ins) for bi, b in enumerate(blocks) for ii, ins in enumerate(b["insns"]) if ins.is_marker]
    stats = {"hidden_loads": len(hidden), "consume_markers": len(markers), "counted_waits": []}
    if not hidden and not markers:
        return viol, stats
    all_dst = set()
    for _bi, _ii, ins in hidden:
        all_dst |= ins.hidden_dst
    # R2
    for _bi, _ii, mk in markers:
        extra = mk.marker_regs - all_dst
        if extra:
            viol.append(f"{name}: line {mk.line}: the marker names v{sorted(extra)} which no hidden load of this kernel "
                        f"writes -- the value was copied between its load and the wait")
    consumed_somewhere = set()
    for _bi, _ii, mk in markers:
        consumed_somewhere |= mk.marker_regs
    for _bi, _ii, ld in hidden:
        if not ld.hidden_dst <= consumed_somewhere:
            viol.append(f"{name}: line {ld.line}: hidden load `{ld.text}` is never named by a NTLINT_CONSUME marker")
    # R1: forward walk from every hidden load
    for bi, ii, ld in hidden:
        dst = ld.hidden_dst
        seen = set()
        work = [(bi, ii + 1)]
        while work:
            b, start = work.pop()
            if (b, start) in seen:
                continue
            seen.add((b, start))
            stop = False
            for ins in blocks[b]["insns"][start:]:
                if ins.is_marker and dst <= ins.marker_regs:
                    stop = True
                    break
                mw = _VMCNT.match(ins.text) or _VMCNT_ALL.match(ins.text)
                if mw and int(mw.group(1), 0) == 0:
                    stop = True  # everything has landed, whoever asked for it
                    break
                if ins is ld:
                    # around the loop back to the same load without a consume: the previous data were never waited for
                    viol.append(f"{name}: line {ld.line}: hidden load `{ld.text}` can be re-issued before it is consumed")
                    stop = True
                    break
                if ins.regs & dst and not ins.is_marker:
                    viol.append(f"{name}: line {ins.line}: `{ins.text}` touches v{sorted(ins.regs & dst)} while the hidden "
                                f"load of line {ld.line} may still be in flight")
                    stop = True
                    break
                if ins.is_marker and ins.marker_regs & dst:
                    viol.append(f"{name}: line {ins.line}: marker names only part of the load of line {ld.line}")
                    stop = True
                    break
            if not stop:
                for s in blocks[b]["succ"]:
                    work.append((s, 0))
    # R3: backward walk from every marker
    pred = [[] for _ in blocks]
    for i, b in enumerate(blocks):
        for s in b["succ"]:
            pred[s].append(i)
    for bi, ii, mk in markers:
        seen = set()
        work = [(bi, ii)]
        while work:
            b, end = work.pop()
            if (b, end) in seen:
                continue
            seen.add((b, end))
            found = False
            for ins in reversed(blocks[b]["insns"][:end]):
                m = _VMCNT.match(ins.text)
                m_all = _VMCNT_ALL.match(ins.text)
                if ins.wrapped:
                    found = True
                    break
                if m and (ins.in_asm or int(m.group(1)) == 0):
                    if int(m.group(1)) != 0:
                        stats["counted_waits"].append({"line": ins.line, "vmcnt": int(m.group(1))})
                    found = True
                    break
                if m_all and int(m_all.group(1), 0) == 0:
                    found = True
                    break
                if ins.is_marker and ins is not mk:
                    found = True  # behind an earlier marker of the same wait (two marker statements in a row)
                    break
                if _VMEM.match(ins.text):
                    viol.append(f"{name}: line {mk.line}: the marker is reached from `{ins.text}` (line {ins.line}) "
                                f"without an inline s_waitcnt vmcnt in between")
                    found = True
                    break
            if not found:
                if not pred[b] and b == 0:
                    viol.append(f"{name}: line {mk.line}: a path from the kernel entry reaches the marker without a wait")
                for p in pred[b]:
                    work.append((p, len(blocks[p]["insns"])))
    # one entry per distinct counted wait
    uniq = {(w["line"], w["vmcnt"]) for w in stats["counted_waits"]}
    stats["counted_waits"] = [{"line": l, "vmcnt": n} for l, n in sorted(uniq)]
    return viol, stats


def lint_text(asm_text):
    """-> {"kernels": {name: stats}, "violations": [...], "sites": n}"""
    funcs = parse_functions(asm_text)
    report = {"kernels": {}, "violations": [], "hidden_loads": 0, "consume_markers": 0}
    for name, blocks in funcs.items():
        viol, stats = lint_function(name, blocks)
        if stats["hidden_loads"] or stats["consume_markers"]:
            report["kernels"][name] = stats
            report["hidden_loads"] += stats["hidden_loads"]
            report["consume_markers"] += stats["consume_markers"]
        report["violations"] += viol
    return report


def lint_file(path):
    with open(path, errors="replace") as f:
        return lint_text(f.read())


if __name__ == "__main__":
    bad = 0
    for p in sys.argv[1:]:
        rep = lint_file(p)
        print(json.dumps({"file": p, "kernels_with_sites": len(rep["kernels"]), "hidden_loads": rep["hidden_loads"],
                          "consume_markers": rep["consume_markers"], "violations": rep["violations"][:20]}, indent=1))
        bad += len(rep["violations"])
    sys.exit(1 if bad else 0)
