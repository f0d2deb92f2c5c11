"""Build profiles/<name>_traffic_pmc.json from two rocprofv3 counter-collection csv files (one --pmc FETCH_SIZE pass, one
--pmc WRITE_SIZE pass of the same bench.py command).
usage: pmc_traffic.py <fetch.csv> <write.csv> <out.json> [<bench line json of the same command>]
The bench line supplies the workload / batch, and the source fingerprint (bench.sources_sha) ties the summary to the
sources it was measured at: bench.py quotes `roofline.traffic` only from a summary whose three keys match its own run.
Per-launch HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024, the gfx950 correction of MI355X_MICROARCH.md (HBM section)."""
import collections, csv, json, os, re, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def label_of(kernel_name: str):
    """bench.py's label of a timed launch (nerve_cl/_nvq.py) for a rocprofv3 kernel name, or None"""
    m = re.search(r"(conv_bf16_kernel|conv_f32_kernel)<(\d+), (\d+)(?:, (?:true|false), (\d+), (\d+))?", kernel_name)
    if m:
        # conv_bf16_kernel<NB, KS, INB, NW, CS>: a workgroup computes NB * CS blocks of 16 output channels, which is what
        # bench.py's label counts (the channel-split 64-channel kernel is <2, 3, true, 8, 2> = label <4,3>)
        nb = int(m.group(2)) * (int(m.group(5)) if m.group(5) else 1)
        return f"{m.group(1)}<{nb},{m.group(3)}>"
    # the 32x32x16-MFMA forms of the same launches (conv_m32.hip, wgrad_m32.hip) carry the label of the descriptor they serve:
    # conv_m32_kernel = 3x3, cout <= 32; wgrad_m32_kernel = 3x3 all-input-channel; wgrad1_m32_kernel = 1x1
    if "conv_m32_kernel<" in kernel_name:
        return "conv_bf16_kernel<2,3>"
    if "wgrad1_m32_kernel<" in kernel_name:
        return "wgrad_bf16_kernel<1>"
    if "wgrad_m32_kernel<" in kernel_name:
        return "wgrad_bf16_kernel<3>"
    m = re.search(r"(wgrad_bf16_kernel|wgrad_f32_kernel)<(\d+)", kernel_name)
    if m:
        return f"{m.group(1)}<{m.group(2)}>"
    return "rdb_tail_kernel" if ("rdb_tail_kernel" in kernel_name or "rdb_tail8_kernel" in kernel_name) else None


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    names = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        lab = label_of(r["Kernel_Name"])
        if lab:
            acc[lab].append(float(r["Counter_Value"]))
            names.setdefault(lab, set()).add(re.sub(r"\(.*", "", r["Kernel_Name"].replace("void nvq::", "")))
    return acc, names


fetch, names = per_kernel(sys.argv[1], "FETCH_SIZE")
write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two separate passes) --output-format csv -- python3 "
                  "bench.py <workload flags> --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timer (tools/r2_measure.sh)",
       "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE counts 64 B per 128-B request on gfx950 "
                     "(MI355X_MICROARCH.md, HBM)", "kernels": {}}
for lab in fetch:
    f = sum(fetch[lab]) / len(fetch[lab])
    w = sum(write[lab]) / max(len(write[lab]), 1)
    out["kernels"][lab] = {"launches": len(fetch[lab]), "fetch_size_kb_per_launch": f, "write_size_kb_per_launch": w,
                           "hbm_bytes_per_launch": (2 * f + w) * 1024, "rocprof_names": sorted(names[lab])}
import bench  # noqa: E402
out["sources_sha"] = bench.sources_sha()
if len(sys.argv) > 4:
    line = json.loads([l for l in open(sys.argv[4]) if l.startswith("{")][-1])
    out["workload"], out["batch"] = line["config"]["workload"], line["config"]["per_gpu_batch"]
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e9, 3) for k, v in out["kernels"].items()}))
