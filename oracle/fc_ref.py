"""CPU oracle (test infrastructure only) for the fused fc.1 epilogue: Linear(50176, 4096) -> LeakyReLU(0.1) ->
Dropout(0.2), yolo_toy_regressor_torch.py:258-263.  numpy restatement of what torch does there, with the two
build-defined pieces spelled out: the split-K slabs are summed in slab order starting from the bias, and the dropout
uniforms come from Philox4x32-10 (key = seed, counter = element index / 4) instead of torch's generator, so the mask is
reproducible outside torch.  With p_drop = 0 the forward equals F.leaky_relu(partial.sum(0) + bias) rounded to bf16 —
that case is checked against torch itself in tests/; the dropout mask has no reference counterpart to pin against
(torch's own mask depends on its generator state): "parity unpinned" for the mask, pinned for everything else."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(seed, ctr):
    """ctr: uint64 array of counters -> uint32 array [len, 4]."""
    ctr = np.asarray(ctr, np.uint64)
    c = [(ctr & np.uint64(0xFFFFFFFF)).astype(np.uint32), (ctr >> np.uint64(32)).astype(np.uint32),
         np.zeros(ctr.shape, np.uint32), np.zeros(ctr.shape, np.uint32)]
    k0, k1 = np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c[0].astype(np.uint64)
            p1 = M1 * c[2].astype(np.uint64)
            h0, l0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & np.uint64(0xFFFFFFFF)).astype(np.uint32)
            h1, l1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & np.uint64(0xFFFFFFFF)).astype(np.uint32)
            c = [h1 ^ c[1] ^ k0, l1, h0 ^ c[3] ^ k1, l0]
            k0 = np.uint32(k0 + W0)
            k1 = np.uint32(k1 + W1)
    return np.stack(c, axis=-1)


def to_bf16(x):
    """fp32 -> bf16 (round to nearest even), returned as fp32 holding the rounded value."""
    u = np.asarray(x, np.float32).view(np.uint32)
    r = ((u >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)
    with np.errstate(over="ignore"):
        out = ((u + r) & np.uint32(0xFFFF0000)).astype(np.uint32)
    nan = np.isnan(x)
    out = np.where(nan, u | np.uint32(0x00400000), out).astype(np.uint32)
    return out.view(np.float32)


def fc_epilogue_fwd(partial, bias, slope=0.1, p_drop=0.0, seed=0):
    """partial fp32 [S, M, N] -> (y [M, N] fp32 holding bf16 values, keep uint8 [M, N])."""
    partial = np.asarray(partial, np.float32)
    s, m, n = partial.shape
    v = np.broadcast_to(np.asarray(bias, np.float32), (m, n)).copy() if bias is not None else np.zeros((m, n), np.float32)
    for k in range(s):
        v = (v + partial[k]).astype(np.float32)
    t = np.where(v > 0, v, (v * np.float32(slope)).astype(np.float32)).astype(np.float32)
    keep = np.ones((m, n), np.uint8)
    if p_drop > 0:
        r = philox4x32_10(seed, np.arange(m * n // 4, dtype=np.uint64)).reshape(m, n)
        u = (r >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
        keep = (u >= np.float32(p_drop)).astype(np.uint8)
        scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p_drop))
        t = np.where(keep == 1, (t * scale).astype(np.float32), np.float32(0.0)).astype(np.float32)
    return to_bf16(t), keep


def fc_epilogue_bwd(dy, y, keep, slope=0.1, p_drop=0.0):
    """dy, y: fp32 arrays holding bf16 values [M, N] -> (dz as bf16-valued fp32, db fp32 [N])."""
    t = np.asarray(dy, np.float32).copy()
    if p_drop > 0:
        scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p_drop))
        t = np.where(keep == 1, (t * scale).astype(np.float32), np.float32(0.0)).astype(np.float32)
    t = np.where(np.asarray(y, np.float32) > 0, t, (t * np.float32(slope)).astype(np.float32)).astype(np.float32)
    dz = to_bf16(t)
    return dz, dz.sum(axis=0, dtype=np.float64).astype(np.float32)
