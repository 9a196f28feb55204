class CallbackList:
    """Fan-out container, the only tensorflow symbol the reference uses."""

    def __init__(self, callbacks=None):
        self.callbacks = list(callbacks or [])

    def set_model(self, model):
        self.model = model
        for c in self.callbacks:
            c.set_model(model)

    def _fan(self, hook, *args):
        for c in self.callbacks:
            getattr(c, hook)(*args)

    def on_train_begin(self, logs=None):
        self._fan("on_train_begin", logs)

    def on_train_end(self, logs=None):
        self._fan("on_train_end", logs)

    def on_epoch_begin(self, epoch, logs=None):
        self._fan("on_epoch_begin", epoch, logs)

    def on_epoch_end(self, epoch, logs=None):
        self._fan("on_epoch_end", epoch, logs)
