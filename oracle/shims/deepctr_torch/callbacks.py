class History:
    """Keras-style record of per-epoch logs."""

    def set_model(self, model):
        self.model = model

    def on_train_begin(self, logs=None):
        self.epoch, self.history = [], {}

    def on_epoch_begin(self, epoch, logs=None):
        pass

    def on_epoch_end(self, epoch, logs=None):
        self.epoch.append(epoch)
        for k, v in (logs or {}).items():
            self.history.setdefault(k, []).append(v)

    def on_train_end(self, logs=None):
        pass
