"""keras.callbacks.ModelCheckpoint and keras.callbacks.EarlyStopping as the reference constructs them (cnn.py:143-144):

    callbacks = [keras.callbacks.ModelCheckpoint(filepath, monitor='val_loss', verbose=0, save_best_only=True, mode='auto'),
                 keras.callbacks.EarlyStopping(monitor='val_loss', patience=5, verbose=0, mode='auto')]

so that `model.fit(..., callbacks=[...])` reads here as it reads there.  They are descriptions -- `Trainer.fit` runs the loop
and consults the two small state machines below, which restate Keras 2.4's rules (tensorflow/python/keras/callbacks.py):
  ModelCheckpoint   on_epoch_end: save when `current < best` (save_best_only) or always; `best` starts at +inf.
  EarlyStopping     on_epoch_end: `current - min_delta < best` resets `wait` (and remembers the weights if
                    restore_best_weights), otherwise `wait += 1` and training stops once `wait >= patience`.
Only what a loss needs is built: monitor 'val_loss' (or 'loss'), mode 'auto' / 'min'."""
from __future__ import annotations

import math
from typing import Optional


def _check_monitor(monitor: str, mode: str) -> None:
    if monitor not in ("val_loss", "loss"):
        raise ValueError(f"monitor={monitor!r}: the training path reports 'loss' and 'val_loss' only")
    if mode not in ("auto", "min"):
        raise ValueError(f"mode={mode!r}: a loss is minimised ('auto' or 'min')")


class ModelCheckpoint:
    def __init__(self, filepath: str, monitor: str = "val_loss", verbose: int = 0, save_best_only: bool = False,
                 save_weights_only: bool = False, mode: str = "auto", save_freq="epoch"):
        _check_monitor(monitor, mode)
        if save_weights_only:
            raise ValueError("save_weights_only=True is not built: the reference saves full models (cnn.py:143)")
        if save_freq != "epoch":
            raise ValueError("save_freq: only 'epoch'")
        self.filepath, self.monitor, self.verbose, self.save_best_only = str(filepath), monitor, int(verbose), bool(save_best_only)
        self.best = math.inf

    def reset(self) -> None:
        self.best = math.inf

    def should_save(self, current: Optional[float]) -> bool:
        """Keras: without save_best_only every epoch is saved; with it, an epoch whose monitored value is missing is skipped
        (a warning there) and one that is not below `best` is not saved.  NaN never compares below."""
        if not self.save_best_only:
            return True
        if current is None:
            return False
        if current < self.best:
            self.best = current
            return True
        return False


class EarlyStopping:
    def __init__(self, monitor: str = "val_loss", min_delta: float = 0.0, patience: int = 0, verbose: int = 0, mode: str = "auto",
                 baseline: Optional[float] = None, restore_best_weights: bool = False):
        _check_monitor(monitor, mode)
        if patience < 0:
            raise ValueError("patience must be >= 0")
        self.monitor, self.min_delta, self.patience, self.verbose = monitor, abs(float(min_delta)), int(patience), int(verbose)
        self.baseline, self.restore_best_weights = baseline, bool(restore_best_weights)
        self.reset()

    def reset(self) -> None:
        self.wait = 0
        self.stopped_epoch = 0
        self.best = math.inf if self.baseline is None else float(self.baseline)

    def update(self, current: Optional[float]):
        """-> (improved, stop) for this epoch's monitored value."""
        if current is None:
            return False, False
        if current - self.min_delta < self.best:
            self.best = current
            self.wait = 0
            return True, False
        self.wait += 1
        return False, self.wait >= self.patience
