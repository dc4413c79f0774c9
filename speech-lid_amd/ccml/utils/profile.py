"""Cost recorder (reference API: ccml/utils/profile.py:8-68) — wall-clock totals per key, plus optional HIP-event timing so
GPU sections are measured on the stream instead of by un-synchronised host clocks."""
import functools
import logging
import threading
import time
from collections import defaultdict


class TimeCostRecoder:
    _lock = threading.Lock()
    _instance = None

    def __new__(cls):
        with cls._lock:
            if cls._instance is None:
                cls._instance = super().__new__(cls)
                cls._instance.values_map = defaultdict(float)
                cls._instance.count_map = defaultdict(int)
                cls._instance._pending = []
        return cls._instance

    def recoder(self, key: str, duration: float):
        self.values_map[key] += duration
        self.count_map[key] += 1

    def gpu_section(self, key: str):
        """Context manager timing a section with HIP events on the current stream (resolved lazily at print time)."""
        return _GpuSection(self, key)

    def _drain(self):
        for key, a, b in self._pending:
            b.synchronize()
            self.recoder(key, a.elapsed_time(b) / 1e3)
        self._pending.clear()

    def format_print(self):
        self._drain()
        for key, total in sorted(self.values_map.items(), key=lambda kv: kv[1], reverse=True):
            n = max(self.count_map[key], 1)
            logging.info("cost %-24s total %.2fs  avg %.2fms  count %d", key, total, total / n * 1e3, self.count_map[key])
        self._clear()

    def _clear(self):
        self.values_map = defaultdict(float)
        self.count_map = defaultdict(int)


class _GpuSection:
    def __init__(self, rec, key):
        self.rec, self.key = rec, key

    def __enter__(self):
        import torch
        self.a, self.b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.a.record()

    def __exit__(self, *exc):
        self.b.record()
        self.rec._pending.append((self.key, self.a, self.b))


_time_cost_recoder = TimeCostRecoder()


def register_cost_statistic(need_return: bool = False):
    def decorator(func):
        key = func.__name__

        @functools.wraps(func)
        def wrapper(*args, **kwargs):
            t0 = time.time()
            out = func(*args, **kwargs)
            _time_cost_recoder.recoder(key, time.time() - t0)
            return out if need_return else None
        return wrapper
    return decorator
