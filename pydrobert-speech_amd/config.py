"""Package constants (reference: config.py:27-52)

`LOG_FLOOR_VALUE` is read when an STFT plan is created (the reference reads the module
global at call time, compute.py:397,459; a plan snapshots it -- change the value, then
build a new computer).  `EFFECTIVE_SUPPORT_THRESHOLD` is read when a bank is built.
"""

__all__ = ["EFFECTIVE_SUPPORT_THRESHOLD", "FLOAT64_ARITHMETIC", "HOST_FEED", "LOG_FLOOR_VALUE", "USE_FFTPACK"]

#: kept for API compatibility; the DFT is always computed on the GPU here
USE_FFTPACK: bool = False

#: magnitude below which a filter response counts as zero (reference config.py:43)
EFFECTIVE_SUPPORT_THRESHOLD: float = 5e-4

#: floor applied before every logarithm (reference config.py:52)
LOG_FLOOR_VALUE: float = 1e-5

#: How float64 signals are computed by the STFT computer.  ``"float64"`` (default): float64
#: arithmetic throughout, the reference's own internal precision, on the generic kernels
#: (results within 1e-9 of the reference; ~370 M frames/s at power-of-two transform sizes).  ``"float32"``: the fused
#: kernel rounds every sample to float32 as it loads the frame and widens the features at the
#: store (results within the float32 tolerance 1e-5 + 1e-4 |ref|; no conversion pass over the
#: signal at transform sizes 256 ... 2048), so the returned dtype still follows the input as in
#: the reference (compute.py:601); an `out` tensor of float32 receives float32 features, the dtype
#: flow of the reference's drivers (command_line.py:107-108, 345-350).  Read at every launch.
FLOAT64_ARITHMETIC: str = "float64"

#: ragged packed batches (less than 0.9 of the utterance x longest-frame-count grid exists) take the launch whose
#: waves walk contiguous stretches of the existing frames (``pds_stft_batch_ragged_f32``) -- same values, +9 %
#: on lengths uniform in 1 ... 15 s.  ``False`` (or PDS_RAGGED_SCHEDULING=0 in the environment) keeps the plain launch (A/B runs).
RAGGED_SCHEDULING = __import__("os").environ.get("PDS_RAGGED_SCHEDULING", "1") != "0"

#: ``compute_full_batch`` of HOST signals (numpy arrays) goes through the pinned staging ring of ``feed.HostFeed``
#: (slot-sized pieces of the batch upload, compute and download concurrently: 3 x the rate of one concatenate and one
#: pageable copy each way) once the batch holds a few seconds of audio.  The ring keeps ~0.5 GB of pinned host
#: memory per computer and sample dtype.  ``False`` (or PDS_HOST_FEED=0): always the plain path.
HOST_FEED = __import__("os").environ.get("PDS_HOST_FEED", "1") != "0"
