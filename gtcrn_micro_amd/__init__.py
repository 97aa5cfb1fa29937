"""gtcrn_micro_amd -- MI355X (gfx950) native hot path of GTCRN-Micro.

Layout mirrors the reference package for the path it replaces:
  gtcrn_micro_amd.models.gtcrn_micro.GTCRNMicro               <- gtcrn_micro.models.gtcrn_micro.GTCRNMicro
  gtcrn_micro_amd.streaming.gtcrn_micro_stream.StreamGTCRNMicro <- gtcrn_micro.streaming.gtcrn_micro_stream
  gtcrn_micro_amd.streaming.conversion.convert.convert_to_stream
The arithmetic lives in csrc/ (HIP, built into libgtcrn_micro_hip.so) behind the C ABI of
include/gtcrn_micro_hip.h; nothing here falls back to the CPU.
"""
from ._lib import f32_to_pcm16, pcm16_to_f32, Engine, GtcrnError, Trainer, istft, make_window, num_frames, selftest_mfma, selftest_split3, stft, stft_frames  # noqa: F401

__all__ = ["pcm16_to_f32", "f32_to_pcm16", "Engine", "GtcrnError", "Trainer", "stft", "istft", "stft_frames", "make_window", "num_frames", "selftest_mfma", "selftest_split3"]
