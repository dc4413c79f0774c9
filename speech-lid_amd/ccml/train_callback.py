class Callback:
    """Base class of trainer callbacks (reference: ccml/train_callback.py:6-40).  Each hook receives one ``value`` dict."""

    def __init__(self, interval: int = 1, *args, **kwargs):
        self.trainer = None
        self.interval = interval
        self.after_eval_epoch_count = 0
        self.after_eval_loop_count = 0
        self.after_train_epoch_count = 0
        self.after_train_loop_count = 0

    def add_trainer(self, trainer):
        self.trainer = trainer

    def before_train_epoch(self, *args, **kwargs):
        pass

    def after_train_loop(self, *args, **kwargs):
        pass

    def after_train_epoch(self, *args, **kwargs):
        pass

    def after_eval_loop(self, *args, **kwargs):
        pass

    def after_eval_epoch(self, *args, **kwargs):
        pass

    def test_loop_end(self, *args, **kwargs):
        pass
