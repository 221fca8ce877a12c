"""Host-side step pieces: quality log record format and sidecar rule of
utils/logger_quality.py:51-113 of the reference, LR schedule golden values.  CPU only."""
import json

from yolo_from_scratch_amd import run


def test_log_quality_record_format(tmp_path):
    log = tmp_path / "q.log"
    rec = run.log_quality(3, 0.01, 1.5, 1.25, 0.42, log_path=log, side_dir=tmp_path)
    lines = log.read_text().splitlines()
    assert len(lines) == 1 and " " not in lines[0]                      # compact separators, one record per line
    got = json.loads(lines[0])
    assert list(got.keys()) == ["tag", "date", "commit", "epoch", "lr", "train", "val", "val_iou"]   # reference order
    assert got == rec and got["tag"] == "QUALITY" and got["epoch"] == 3 and got["val_iou"] == 0.42
    run.log_quality(4, 0.01, 1.4, 1.2, 0.5, best=True, log_path=log, side_dir=tmp_path, extra={"images_per_s": 4000.0})
    last = json.loads(log.read_text().splitlines()[-1])
    assert last["tag"] == "QUALITY_BEST" and last["images_per_s"] == 4000.0
    if last["commit"] != "NA":
        side = json.loads((tmp_path / last["commit"] / "best.json").read_text())
        assert side["epoch"] == 4
        # a worse record must not replace the sidecar; an equal-IoU, lower-val-loss one must
        run.log_quality(5, 0.01, 1.4, 1.3, 0.45, best=True, log_path=log, side_dir=tmp_path)
        assert json.loads((tmp_path / last["commit"] / "best.json").read_text())["epoch"] == 4
        run.log_quality(6, 0.01, 1.4, 1.1, 0.5, best=True, log_path=log, side_dir=tmp_path)
        assert json.loads((tmp_path / last["commit"] / "best.json").read_text())["epoch"] == 6


def test_is_better_ordering():
    a = {"val_iou": 0.5, "val": 1.0, "epoch": 3}
    assert run.is_better(a, None)
    assert run.is_better({"val_iou": 0.6, "val": 2.0, "epoch": 1}, a)            # IoU first
    assert run.is_better({"val_iou": 0.5, "val": 0.9, "epoch": 1}, a)            # then lower val loss
    assert run.is_better({"val_iou": 0.5, "val": 1.0, "epoch": 4}, a)            # then later epoch
    assert not run.is_better(dict(a), a)                                         # strictly better only


def test_init_head_biases_prior_logits():
    """practice/yolo_toy_regressor_3_torch.py:263-276 on a plain torch head (host code, no kernels)."""
    import math
    import torch
    import torch.nn as nn
    from yolo_from_scratch_amd.anchors import init_head_biases

    class M(nn.Module):
        def __init__(self):
            super().__init__()
            self.fc = nn.Sequential(nn.Flatten(), nn.Linear(16, 8), nn.ReLU(), nn.Linear(8, 4))
    m = M()
    torch.manual_seed(0)
    last = init_head_biases(m, prior_xy=(0.5, 0.4), prior_wh=(0.25, 0.35))
    assert last is m.fc[3]
    exp = [math.log(p / (1 - p)) for p in (0.5, 0.4, 0.25, 0.35)]
    assert torch.allclose(m.fc[3].bias.detach(), torch.tensor(exp, dtype=torch.float32))
    bound = math.sqrt(6.0 / (8 + 4))
    assert float(m.fc[3].weight.abs().max()) <= bound
