"""CometLogger: the reference logs to an external service / package (ccml/loggers/comet_logger.py:18-22 (comet_ml client)) that is outside the hot path (out of
scope).  This class keeps launch scripts and YAML `logger:` blocks importable: it ALWAYS writes to the local JSONL logger - there
is no remote client, no opt-in switch and no credential handling in this repository."""
from ccml.loggers.jsonl_logger import JsonlLogger


class CometLogger(JsonlLogger):
    def __init__(self, *args, project: str = "lid", name: str = "run", **kwargs):
        super().__init__(path=f"{project}-{name}.comet.jsonl", name=name)
