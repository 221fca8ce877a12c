#!/usr/bin/env python3
"""Phase timing of nms_kernel (needs libyfs_dbg.so: `make -C yolo_from_scratch_amd/csrc nms-timing`; see the NMS_T marks in box_kernels.hip).
Prints core-clock cycles between the marks for workgroups 0 and 517 of a 1024-image launch (config 4)."""
import ctypes, os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import yolo_from_scratch_amd as y
from yolo_from_scratch_amd import synth
lib = ctypes.CDLL(os.path.join(root, "yolo_from_scratch_amd", "libyfs_dbg.so"))
N, S, A, C = 1024, 13, 5, 20
raw = torch.from_numpy(synth.raw_head_v2(N, S, A, C, seed=1234)).cuda()
anc = torch.from_numpy(synth.VOC_ANCHORS[:A]).cuda()
boxes, score, cls = y.decode_grid(raw, anc)
nb = S * S * A
boxes = boxes.reshape(N, nb, 4).contiguous(); score = score.reshape(N, nb).contiguous(); cls = cls.reshape(N, nb).int().contiguous()
keep = torch.empty(N, nb, dtype=torch.int32, device="cuda"); count = torch.empty(N, dtype=torch.int32, device="cuda")
P = lambda t: ctypes.c_void_p(t.data_ptr())
for it in range(3):
    rc = lib.yfs_nms_classwise_f32(P(boxes), P(score), P(cls), ctypes.c_int64(N), ctypes.c_int(nb), ctypes.c_float(0.05),
                                   ctypes.c_float(0.5), P(keep), P(count), None)
    assert rc == 0, rc
    torch.cuda.synchronize()
buf = (ctypes.c_longlong * 32)()
assert lib.yfs_debug_nms_times(buf) == 0
names = ["A filter", "pad + sort", "counting sort", "pair prefix", "C1 pair IoUs", "C2 resolve", "D output"]
for b in range(2):
    t = [buf[b * 16 + k] for k in range(8)]
    print("block", (0, 517)[b], "total", t[7] - t[0], "cycles:", ", ".join(f"{names[k]} {t[k + 1] - t[k]}" if k < 7 else "" for k in range(7)))
print("count[0], count[517] =", int(count[0]), int(count[517]))
