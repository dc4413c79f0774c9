from ccml.train_callback import Callback


class LrCallback(Callback):
    """Logs the current learning rate once per epoch (reference: ccml/callbacks/lr_callback.py)."""

    def after_train_epoch(self, value=None):
        lr = self.trainer.optimizer.param_groups[0]["lr"]
        self.trainer.logger.log({"lr": lr}, stage="val", commit=False)
