class BaseLogger:
    """Backend interface (reference: ccml/loggers/base_logger.py:5-49)."""

    def __init__(self, *args, **kwargs):
        pass

    def log(self, data=None, *args, **kwargs):
        raise NotImplementedError

    def watch_model(self, model, *args, **kwargs):
        pass

    def save(self, path):
        pass

    def get_resume_state(self):
        return None, None

    def resume_from(self, checkpoint: dict):
        pass

    def get_checkpoint_by_name(self, name: str, path: str = None):
        return None
