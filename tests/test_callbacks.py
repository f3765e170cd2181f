"""callbacks.ModelCheckpoint / EarlyStopping: the two state machines against Keras 2.4's rules, restated here as the plain
loops of tensorflow/python/keras/callbacks.py (EarlyStopping.on_epoch_end, ModelCheckpoint._save_model); the GPU test runs
cnn.py:135-147 with the reference's own callback list."""
import math
import os

import numpy as np
import pytest

from modulationdetectioncnn_amd.callbacks import EarlyStopping, ModelCheckpoint


def _keras_early_stopping(values, patience, min_delta=0.0, baseline=None):
    """-> (epoch at which training stops or None, epochs that improved)."""
    best = math.inf if baseline is None else baseline
    wait, improved = 0, []
    for ep, cur in enumerate(values):
        if np.less(cur - min_delta, best):
            best, wait = cur, 0
            improved.append(ep)
        else:
            wait += 1
            if wait >= patience:
                return ep, improved
    return None, improved


@pytest.mark.parametrize("patience", [0, 1, 2, 5])
@pytest.mark.parametrize("min_delta", [0.0, 0.05])
def test_early_stopping_rule(patience, min_delta):
    rng = np.random.default_rng(patience * 7 + int(min_delta * 100))
    for trial in range(50):
        values = list(np.round(np.cumsum(rng.normal(-0.02, 0.08, 30)) + 2.0, 3))
        if trial % 5 == 0:
            values[7] = float("nan")                      # NaN never improves (np.less is False)
        es = EarlyStopping(monitor="val_loss", patience=patience, min_delta=min_delta, verbose=0, mode="auto")
        got_stop, got_improved = None, []
        for ep, v in enumerate(values):
            improved, halt = es.update(v)
            if improved:
                got_improved.append(ep)
            if halt:
                got_stop = ep
                break
        assert (got_stop, got_improved) == _keras_early_stopping(values, patience, min_delta)
        es.reset()
        assert es.wait == 0 and es.best == math.inf


def test_early_stopping_baseline_and_missing_monitor():
    es = EarlyStopping(patience=2, baseline=1.0)
    assert [es.update(v) for v in (1.5, 1.2)] == [(False, False), (False, True)]      # never got below the baseline
    es = EarlyStopping(patience=1)
    assert es.update(None) == (False, False)                                          # no validation data: Keras warns and skips


def test_checkpoint_rule():
    c = ModelCheckpoint("f.h5", monitor="val_loss", verbose=0, save_best_only=True, mode="auto")
    assert [c.should_save(v) for v in (2.0, 2.0, 1.5, float("nan"), 1.6, 1.4, None)] == [True, False, True, False, False, True, False]
    every = ModelCheckpoint("f.h5")                                                   # Keras' default: every epoch
    assert all(every.should_save(v) for v in (2.0, 3.0, None))


def test_refusals():
    for bad in (dict(monitor="val_accuracy"), dict(mode="max")):
        with pytest.raises(ValueError):
            EarlyStopping(**bad)
        with pytest.raises(ValueError):
            ModelCheckpoint("f.h5", **bad)
    with pytest.raises(ValueError):
        ModelCheckpoint("f.h5", save_weights_only=True)
    with pytest.raises(ValueError):
        EarlyStopping(patience=-1)


@pytest.mark.gpu
def test_cnn_py_135_147_with_the_references_own_callback_list(tmp_path):
    import modulationdetectioncnn_amd.callbacks as callbacks           # `keras.callbacks` in the reference
    from modulationdetectioncnn_amd import VTCNN2, Topology
    from modulationdetectioncnn_amd.training import to_onehot
    from tests.test_training_gpu import _leveled
    x, lab = _leveled(3000, seed=5)
    X_train, Y_train, X_test, Y_test = x[:2000], to_onehot(lab[:2000], 3), x[2000:], to_onehot(lab[2000:], 3)
    nb_epoch, batch_size = 40, 512

    def run(**fit_kw):
        model = VTCNN2.synthetic(Topology.deployed(3), seed=4, device=0)
        model.compile(loss='categorical_crossentropy', optimizer='adam', lr=0.01)
        history = model.fit(X_train, Y_train, batch_size=batch_size, epochs=nb_epoch, verbose=2, validation_data=(X_test, Y_test),
                            seed=0, **fit_kw)
        return model, history

    filepath = str(tmp_path / 'convmodrecnets_CNN2_0.5.wts.h5')
    model, history = run(callbacks=[
        callbacks.ModelCheckpoint(filepath, monitor='val_loss', verbose=0, save_best_only=True, mode='auto'),
        callbacks.EarlyStopping(monitor='val_loss', patience=5, verbose=0, mode='auto')])
    short = str(tmp_path / 'short.h5')
    model_s, history_s = run(checkpoint=short, patience=5)                # the short form is the same loop, bit for bit
    assert history.history == history_s.history and history.stopped_epoch == history_s.stopped_epoch
    assert open(filepath, 'rb').read() == open(short, 'rb').read()
    model.load_weights(filepath)                                          # cnn.py:147
    score = model.evaluate(X_test, Y_test, verbose=0, batch_size=batch_size)
    assert abs(score - min(history.history['val_loss'])) <= 2e-5 * score
    # restore_best_weights: the model itself ends on the best epoch's weights; save_best_only=False: the file is the LAST epoch's
    last = str(tmp_path / 'last.h5')
    model_r, history_r = run(callbacks=[callbacks.ModelCheckpoint(last), callbacks.EarlyStopping(patience=2, restore_best_weights=True)])
    if history_r.stopped_epoch is not None:
        assert abs(model_r.evaluate(X_test, Y_test) - min(history_r.history['val_loss'])) <= 2e-5 * score
        m_last = VTCNN2.from_h5(last, device=0)
        assert abs(m_last.evaluate(X_test, Y_test) - history_r.history['val_loss'][-1]) <= 2e-5 * score
    with pytest.raises(ValueError):
        run(callbacks=[callbacks.EarlyStopping(patience=1)], checkpoint=short)
    with pytest.raises(ValueError):
        run(callbacks=[object()])
    assert os.path.exists(last)
