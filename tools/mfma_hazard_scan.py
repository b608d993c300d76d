"""Dev tool (build container, no GPU): static scan of the gfx950 ISA of every kernel for the hazard found in round 3 — an instruction
that READS a register of an MFMA's destination closer behind the MFMA than the MFMA's pass count allows, on ANY control-flow path
(labels and branches are followed: the case that bit was a loop exit — last MFMA of the body, branch over the body's tail, compare,
wait, barrier, branch, and a register-allocator copy of the last accumulator register in the exit block, six wait states behind a
16-pass MFMA). The compiler's hazard recogniser pads such reads inside a basic block and on the paths it models; it missed that one.
Usage: hipcc ... -S --cuda-device-only x.hip -o x.s ; python tools/mfma_hazard_scan.py x.s [...]
Wait states between an XDL write and a read of the same VGPR, calibrated on this toolchain (every padded site of the library sits
exactly at these distances): 2-pass 4, 4-pass 6, 8-pass 10, 16-pass 18; an instruction counts 1, `s_nop N` counts N + 1. An MFMA that
takes the result as srcC is exempt (back-to-back accumulation)."""
import re
import sys

PASSES = {"32x32x2_f32": 16, "32x32x2f32": 16, "32x32x16_f16": 8, "32x32x16_bf16": 8, "32x32x8_f16": 16, "16x16x4_f64": 8, "16x16x4f64": 8,
          "32x32x1_f32": 16, "16x16x4_f32": 8, "16x16x16_f16": 8, "4x4x1_f32": 2, "32x32x4_f32": 16, "16x16x1_f32": 8}
NEED = {2: 4, 4: 6, 8: 10, 16: 18}
reg_re = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def regs(tok):
    out = set()
    for m in reg_re.finditer(tok):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def parse(path):
    funcs, cur, labels = [], None, None
    for ln, line in enumerate(open(path), 1):
        s = line.split(";")[0].strip()
        if not s:
            continue
        if s.endswith(":"):
            name = s[:-1]
            if name.startswith(".L"):
                if cur is not None:
                    labels[name] = len(cur["ins"])
            elif not name.startswith("."):
                cur = {"name": name, "ins": [], "labels": {}}
                labels = cur["labels"]
                funcs.append(cur)
            continue
        if s.startswith(".") or cur is None:
            continue
        op = s.split()[0]
        ops = [o.strip() for o in s[len(op):].split(",")]
        ins = {"ln": ln, "text": s, "op": op, "states": 1, "mfma": None, "reads": set(), "target": None, "uncond": False, "end": False}
        if op == "s_nop":
            ins["states"] = int(ops[0]) + 1
        elif op.startswith("v_mfma") or op.startswith("v_smfmac"):
            key = next((k for k in PASSES if k in op), None)
            ins["mfma"] = (regs(ops[0]), NEED[PASSES.get(key, 16)])
            ins["reads"] = set().union(*[regs(o) for o in ops[1:3]])  # srcA / srcB only
        elif op.startswith("s_cbranch") or op == "s_branch":
            ins["target"] = ops[0]
            ins["uncond"] = op == "s_branch"
        elif op in ("s_endpgm", "s_setpc_b64"):
            ins["end"] = True
        elif op[0] in "vdgbf" or op.startswith(("scratch", "global", "flat")):
            r = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
            if op.startswith(("global_store", "ds_write", "scratch_store", "buffer_store", "flat_store", "global_atomic", "ds_add", "ds_max")):
                r |= regs(ops[0])
            if op.startswith(("v_cmp", "v_cmpx")):
                r |= regs(ops[0])
            ins["reads"] = r
        cur["ins"].append(ins)
    return funcs


def scan(path):
    hits = []
    for f in parse(path):
        ins, labels = f["ins"], f["labels"]
        for i, m in enumerate(ins):
            if not m["mfma"]:
                continue
            dst, need = m["mfma"]
            seen = {}
            stack = [(i + 1, need)]
            while stack:
                j, left = stack.pop()
                while j < len(ins) and left > 0:
                    if seen.get(j, 0) >= left:
                        break
                    seen[j] = left
                    x = ins[j]
                    if x["reads"] & dst:
                        hits.append((f["name"], x["ln"], x["text"], m["text"], m["ln"], left))
                    if x["mfma"] and x["mfma"][0] & dst:
                        break  # the destination is rewritten: the newer MFMA is the one that counts from here on
                    left -= x["states"]
                    if x["end"]:
                        break
                    if x["target"] is not None:
                        t = labels.get(x["target"])
                        if t is not None and left > 0:
                            stack.append((t, left))
                        if x["uncond"]:
                            break
                    j += 1
    return sorted(set(hits), key=lambda h: h[1])


if __name__ == "__main__":
    total = 0
    for p in sys.argv[1:]:
        h = scan(p)
        total += len(h)
        for func, ln, s, m, mln, left in h[:20]:
            print(f"{p}:{ln}: {func[:70]}\n    reads the result of line {mln} ({m.split()[0]}) {left} wait state(s) early: {s}")
        print(f"{p}: {len(h)} candidate(s)")
    sys.exit(1 if total else 0)
