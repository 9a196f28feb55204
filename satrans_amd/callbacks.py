"""Keras-style callbacks used by `BaseModel.fit`.

The reference takes `History` from deepctr-torch and `CallbackList` from tensorflow.keras
(models/meta_basemodel.py:21-30,287-294,379-383); neither package is available on the GPU box, and
only their fan-out behaviour is used, so both are restated here.
"""
from __future__ import annotations


class Callback:
    def set_model(self, model):
        self.model = model

    def on_train_begin(self, logs=None):
        pass

    def on_train_end(self, logs=None):
        pass

    def on_epoch_begin(self, epoch, logs=None):
        pass

    def on_epoch_end(self, epoch, logs=None):
        pass


class History(Callback):
    """`history[name]` is the list of per-epoch values; `epoch` the list of epoch indices."""

    def __init__(self):
        self.epoch, self.history = [], {}

    def on_train_begin(self, logs=None):
        self.epoch, self.history = [], {}

    def on_epoch_end(self, epoch, logs=None):
        self.epoch.append(epoch)
        for key, value in (logs or {}).items():
            self.history.setdefault(key, []).append(value)


class CallbackList:
    def __init__(self, callbacks=None):
        self.callbacks = list(callbacks or [])
        self.model = None

    def set_model(self, model):
        self.model = model
        for cb in self.callbacks:
            cb.set_model(model)

    def on_train_begin(self, logs=None):
        for cb in self.callbacks:
            cb.on_train_begin(logs)

    def on_train_end(self, logs=None):
        for cb in self.callbacks:
            cb.on_train_end(logs)

    def on_epoch_begin(self, epoch, logs=None):
        for cb in self.callbacks:
            cb.on_epoch_begin(epoch, logs)

    def on_epoch_end(self, epoch, logs=None):
        for cb in self.callbacks:
            cb.on_epoch_end(epoch, logs)
