"""Summarise the memory / wait / MFMA event order of kernels in a hipcc -S listing: the check that nothing between a prefetch
and its commit waits on vmcnt (a `s_waitcnt vmcnt(0)` between the loads and the MFMA section = the prefetch is exposed).
usage: python tools/isa_events.py file.s substring-of-kernel-symbol [...]"""
import re, sys

def kernels(path):
    name, body = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
        elif name is not None:
            body.append(line)
            if "s_endpgm" in line:
                yield name, body
                name = None

def events(body):
    out, run = [], 0
    for i, line in enumerate(body):
        t = line.split(";")[0].strip()
        if not t:
            continue
        op = t.split()[0]
        if op.startswith("v_mfma"):
            run += 1
            continue
        key = None
        if op.startswith(("global_load", "buffer_load")): key = "LOAD"
        elif op.startswith(("global_store", "buffer_store")): key = "STORE"
        elif op == "s_waitcnt" and "vmcnt" in t: key = t.replace("s_waitcnt ", "")
        elif op == "s_barrier": key = "BARRIER"
        elif op.startswith("scratch_"): key = "SCRATCH!"
        elif t.startswith(".LBB"): key = t.split(":")[0]
        elif op.startswith("s_cbranch") or op == "s_branch": key = t
        if key is None:
            continue
        if run:
            out.append(f"mfma x{run}"); run = 0
        if out and out[-1].split(" x")[0] == key and key in ("LOAD", "STORE"):
            n = int(out[-1].split(" x")[1]) if " x" in out[-1] else 1
            out[-1] = f"{key} x{n + 1}"
        else:
            out.append(key)
    if run: out.append(f"mfma x{run}")
    return out

if __name__ == "__main__":
    path, pats = sys.argv[1], sys.argv[2:]
    for name, body in kernels(path):
        if any(p in name for p in pats):
            print("==", name)
            print("  " + "\n  ".join(events(body)))
