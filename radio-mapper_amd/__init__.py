"""MI355X-native TDoA cross-correlation engine (drop-in behind tdoa_processor.py's API).

  synth           synthetic IQ windows of the BASELINE shapes (numpy)
  xcorr           ctypes binding of the C-ABI library  include/rmx.h  (HIP, gfx950)
  tdoa_processor  host-side mirror of the reference's tdoa_processor.py interface
  multi           the same engine over several devices: one ctx + host thread per GPU, windows block-sharded
  iq_wire         IQ windows on the wire: the reference's JSON complex strings, and a binary frame (host side only)
  shard           window blocks per rank / device, host-side gather (no collective on the data path)
"""
from . import synth  # noqa: F401

__all__ = ["synth"]
