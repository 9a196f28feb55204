"""The step's helper streams, ONE set per device for the whole process.  HIP maps every stream it creates onto one of
GPU_MAX_HW_QUEUES hardware queues in creation order, and streams that share a queue serialise: a second engine that created
streams of its own found them on the queues of the first one's - its every launch waited ~30 us behind packets that were not
its business (tools/fake_world.py with several rank counts in one process: every second run 30 % slower, whatever its N)."""
from typing import Dict

_STREAMS: Dict[tuple, "object"] = {}


def shared_stream(dev, kind: str, make):
    key = (str(dev), kind)
    if key not in _STREAMS:
        _STREAMS[key] = make()
    return _STREAMS[key]
