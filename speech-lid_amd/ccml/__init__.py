"""ccml — drop-in mirror of the reference's mini training framework (ccml/ in kouyt5/speech-lid), rebuilt around the
lidk HIP engine.  Public names, constructor arguments and hook semantics follow the reference so that ``lid/main.py``-style
launchers work unchanged; the implementation is new."""
from ccml.train_helper import seed_everything

__all__ = ["seed_everything"]
