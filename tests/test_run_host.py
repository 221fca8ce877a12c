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
