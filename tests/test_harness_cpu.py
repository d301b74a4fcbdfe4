"""CPU: the epoch-loop bookkeeping of gelslim_depth_amd/harness.py (early stopping on the smoothed validation loss,
best-validation and per-epoch checkpoints, log text) against oracle/harness_ref.py, a literal restatement of
train_utils/train_unet.py:312-523.  Device passes are replaced by host stubs: no GPU, no libgsd compute call."""
import numpy as np
import pytest

from gelslim_depth_amd import harness
from oracle import harness_ref


def _sequences(kind, n=60, seed=0):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    if kind == "u_shape":            # falls, then rises: the stopping rule must fire
        val = 1.0 / (1 + t) + 0.002 * np.maximum(t - 12, 0) ** 1.5
    elif kind == "noisy":
        val = 0.5 * np.exp(-t / 10.0) + 0.05 + 0.02 * rng.standard_normal(n)
    else:                            # plateau with ties
        val = np.maximum(0.3 - 0.02 * t, 0.1)
    train = val * 0.9 + 0.01
    test = val * 1.1
    return train.tolist(), val.tolist(), test.tolist()


@pytest.mark.parametrize("kind", ["u_shape", "noisy", "plateau"])
@pytest.mark.parametrize("indefinitely,save_at", [(False, ()), (True, (3, 7))])
def test_fit_bookkeeping_equals_reference_loop(tmp_path, kind, indefinitely, save_at):
    train, val, test = _sequences(kind)
    max_epochs = 40 if indefinitely else None
    exp_lines, exp_saves, exp_epochs = harness_ref.run(train, val, test, train_indefinitely=indefinitely, save_at_epochs=save_at,
                                                       weights_name="unet_x", max_epochs=max_epochs)
    state = {"e": 0, "phase": 0}
    saved, lines = [], []

    def train_pass(step, loader):
        return train[state["e"]] * 7, 7              # (sum of batch losses, batches)

    def eval_pass(step, loader):
        v = val[state["e"]] if loader == "val" else test[state["e"]]
        if loader == "test":
            state["e"] += 1
        return v

    def save(step, path):
        saved.append(path.split("/")[-1])
        open(path, "w").write("x")
    log = tmp_path / "loss.txt"
    H = harness.fit(None, "train", "val", "test", str(tmp_path / "weights"), "unet_x", loss_values_path=str(log),
                    train_indefinitely=indefinitely, save_at_epochs=save_at, max_epochs=max_epochs, train_pass=train_pass,
                    eval_pass=eval_pass, save=save, echo=lines.append)
    got = [l for l in lines if not l.startswith("Time for epoch") and not l.startswith("Training time")]
    assert got == exp_lines
    assert saved == exp_saves
    assert len(H["validation_loss"]) == exp_epochs and np.allclose(H["validation_loss"], val[:exp_epochs])
    assert np.allclose(H["train_loss"], train[:exp_epochs])
    written = [l for l in log.read_text().splitlines() if not l.startswith("Time for epoch") and not l.startswith("Training time")]
    assert written == exp_lines
    if not indefinitely:
        assert exp_epochs < 60, "the stopping rule never fired on this sequence"


def test_early_stopping_counts_the_zero_initialised_window():
    """The reference's ring starts at zeros, so the smoothed loss RISES for the first `window` epochs even while the raw
    loss falls: with the default threshold 5 a run stops at epoch 7 whatever the losses do -- reproduced, not 'fixed'."""
    es = harness.EarlyStopping(window=10, count_threshold=5)
    stops = [es.update(1.0 / (k + 1))[0] for k in range(8)]
    assert stops.index(True) == 5          # sixth consecutive rise of the mean
