#!/usr/bin/env python3
"""HBM traffic per launch of the HBM-bound kernel sets from two rocprofv3 --pmc passes (tools/evidence.sh):
    python tools/pmc_traffic.py gpurun_out/ev_<tag>  > profiles/rNN/pmc_traffic_hbm_kernels.txt
FETCH_SIZE / WRITE_SIZE are in KB per dispatch.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE
counts a 16-B-per-lane streaming read at half its bytes, so it is doubled; WRITE_SIZE is exact."""
import collections
import csv
import re
import sys

root = sys.argv[1]
ALG = {  # kernel -> (workload, algorithmic MB per launch, SURVEY 8(d) per-unit figure)
    "yolov1_loss_kernel": ("N=65536, S=7 B=2 C=20", 65536 * 17640 / 1e6, "17,640 B/img"),
    "decode_grid_kernel": ("N=1024, 845 boxes, C=20 (decode launch)", None, ""),
    "nms_kernel": ("N=1024, 845 boxes, C=20 (NMS launch)", None, ""),
    "assign_v2_kernel": ("N=8192, S=13 A=5 G=32", 8192 * 32750 / 1e6, "32,750 B/img"),
}


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0].split("<")[0]


val = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{root}/pmc_{c}/hbm_counter_collection.csv")):
        if r["Counter_Name"] == c:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    val[c] = {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}

print("HBM traffic of the HBM-bound kernel sets, rocprofv3 --pmc, separate passes (FETCH_SIZE alone, WRITE_SIZE alone)")
print("command: rocprofv3 --pmc <counter> --output-format csv -- python3 tools/hbm_kernels_bench.py   (tools/evidence.sh)")
print("Counter unit: KB per dispatch, mean over the dispatches of a run.  gfx950 correction (MI355X_MICROARCH.md, HBM")
print("section): FETCH_SIZE reports half of the bytes of a wide (16 B/lane) streaming read -> doubled; WRITE_SIZE exact.\n")
print(f"{'kernel':22s} {'workload':42s} {'n':>3s} {'FETCH KB':>10s} {'x2 read MB':>10s} {'write MB':>9s} {'traffic MB':>10s} {'algorithmic MB':>14s}")
tot = {}
for k, (wl, alg, per) in ALG.items():
    if k not in val["FETCH_SIZE"]:
        continue
    f, n = val["FETCH_SIZE"][k]
    w, _ = val["WRITE_SIZE"][k]
    rd, wr = 2 * f / 1e3, w / 1e3
    tot[k] = rd + wr
    a = f"{alg:10.1f} ({per})" if alg else ""
    print(f"{k:22s} {wl:42s} {n:3d} {f:10.1f} {rd:10.1f} {wr:9.1f} {rd + wr:10.1f} {a}")
if "decode_grid_kernel" in tot and "nms_kernel" in tot:
    print(f"\ndecode + NMS per 1024 images: {tot['decode_grid_kernel'] + tot['nms_kernel']:.1f} MB; algorithmic "
          f"{1024 * 108160 / 1e6:.1f} MB (108,160 B/img: raw head in, boxes/score/class/keep out); the NMS launch re-reads "
          f"score, class and the passing boxes that the decode launch wrote (L2/MALL hits do not count as HBM traffic).")
print("#json " + __import__("json").dumps({"nms": round((tot.get("decode_grid_kernel", 0) + tot.get("nms_kernel", 0)) * 1e6),
                                             "loss": round(tot.get("yolov1_loss_kernel", 0) * 1e6),
                                             "assign": round(tot.get("assign_v2_kernel", 0) * 1e6)}))     # bytes per launch; read by bench.py
