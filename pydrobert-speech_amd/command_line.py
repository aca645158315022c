"""``signals-to-torch-feat-dir`` on the GPU (reference command_line.py:337-607).

Same command line and on-disk result as the reference's tool -- a text map of
``<utt_id> <path>`` lines in, one ``<prefix><utt_id><suffix>`` torch file per utterance out
(``FloatTensor (T, F)``), ``--manifest`` resume -- but the work is organised for the device:

* signals are read by a pool of host threads (``--num-workers``) while the GPU works on the
  previous batch; 16-bit PCM stays int16 until a frame is loaded on the device (a quarter of the
  bytes of the reference's float64 arrays over PCIe; the conversion is exact);
* with ``--staging-ring`` a batch goes through the pinned staging ring of ``feed.HostFeed`` where the
  chain allows (an STFT computer with a fused kernel, at most a pre-emphasis in front): upload,
  kernels and download are queued on one stream without a host concatenate or a pageable copy.  Off
  by default: the tool is bound by reading and writing files (~1 ms per utterance against ~10 us of
  device work), and on a corpus of a thousand utterances the ring's pinned buffers cost more to set
  up than they save (tools/driver_rate.py: 0.6-0.9 s without, 1.4-2.1 s with);
* a batch of utterances (``--batch-utts`` / ``--batch-samples``) is packed into one device
  buffer and goes through ONE launch per stage: dither, pre-emphasis (fused into the frame
  loader when it is the last pre-processor), the fused STFT/filter-bank kernel, then each
  post-processor over the packed rows;
* features come back in one transfer per batch and are written with ``torch.save``.

Differences from the reference, by design: arithmetic is float32 unless ``--precision
float64`` (the reference computes in float64 and stores float32); dither noise comes from the
device's counter-based generator (seeded per utterance with ``seed + index``, as the reference
seeds torch per utterance), so dithered features agree statistically, not sample by sample.

    python -m pydrobert_speech_amd.command_line map.txt fbank.json out_dir [options]
"""
import argparse
import concurrent.futures
import json
import os
import sys
from typing import List, Optional, Sequence

import numpy as np

from .alias import alias_factory_subclass_from_arg
from .compute import FrameComputer
from .post import Deltas, PostProcessor, Stack, Standardize
from .pre import Dither, Preemphasize, PreProcessor
from .util import SIGNAL_SOURCES, read_signal

__all__ = ["FeatureDirWriter", "signals_to_torch_feat_dir"]


def _config(text: str):
    """A JSON file name or a JSON string (reference command_line.py:48-66; YAML is not read)"""
    try:
        with open(text) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        pass
    try:
        return json.loads(text)
    except ValueError:
        raise argparse.ArgumentTypeError(f'Unable to parse "{text}" as a JSON file or string')


def _nonneg(text: str) -> int:
    value = int(text)
    if value < 0:
        raise argparse.ArgumentTypeError(f"{text} is not a non-negative integer")
    return value


def _parse(args):
    ap = argparse.ArgumentParser(
        prog="signals-to-torch-feat-dir", description=__doc__,
        formatter_class=argparse.RawDescriptionHelpFormatter,
    )
    ap.add_argument("map", type=argparse.FileType("r"), help="file of '<utterance> <path>' lines")
    ap.add_argument("computer_config", type=_config, nargs="?", default=None,
                    help="JSON file or string configuring a FrameComputer; without it the audio "
                         "itself is stored with shape (S, 1)")
    ap.add_argument("dir", help="output directory (created if missing)")
    ap.add_argument("--channel", type=int, default=-1, help="channel to use; -1 assumes mono")
    ap.add_argument("--preprocess", type=_config, default=tuple(),
                    help="JSON list of PreProcessor configurations, applied in order")
    ap.add_argument("--postprocess", type=_config, default=tuple(),
                    help="JSON list of PostProcessor configurations, applied in order")
    ap.add_argument("--force-as", default=None, choices=sorted(SIGNAL_SOURCES),
                    help="read every path as this type instead of going by its name")
    ap.add_argument("--seed", type=_nonneg, default=None, help="seed for dithering")
    ap.add_argument("--file-prefix", default="")
    ap.add_argument("--file-suffix", default=".pt")
    ap.add_argument("--num-workers", type=_nonneg, default=0,
                    help="host threads reading signals ahead of the GPU (0: read in the main thread)")
    ap.add_argument("--manifest", type=argparse.FileType("a+"), default=None,
                    help="utterances already listed here are skipped; finished ones are appended")
    ap.add_argument("--batch-utts", type=int, default=256, help="utterances per device batch")
    ap.add_argument("--batch-samples", type=int, default=1 << 26, help="samples per device batch")
    ap.add_argument("--precision", choices=("float32", "float64"), default="float32")
    ap.add_argument("--staging-ring", action="store_true",
                    help="send the batches through a pinned staging ring (feed.HostFeed); pays on very large corpora only")
    return ap.parse_args(args)


def _as_list(spec, base):
    if isinstance(spec, dict) or isinstance(spec, str):
        spec = [spec]
    return [alias_factory_subclass_from_arg(base, element) for element in spec]


class FeatureDirWriter:
    """Batches utterances through the device pipeline and writes one feature file each"""

    def __init__(self, computer: Optional[FrameComputer], preprocessors: Sequence[PreProcessor],
                 postprocessors: Sequence[PostProcessor], out_dir: str, channel: int = -1,
                 force_as: Optional[str] = None, seed: int = 0, file_prefix: str = "",
                 file_suffix: str = ".pt", manifest=None, precision: str = "float32", staging_ring: bool = False):
        for pre in preprocessors:
            if not isinstance(pre, (Dither, Preemphasize)):
                raise NotImplementedError(f"pre-processor {type(pre).__name__}")
        self.computer, self.pre, self.post = computer, list(preprocessors), list(postprocessors)
        self.out_dir, self.channel, self.force_as, self.seed = out_dir, channel, force_as, seed
        self.prefix, self.suffix, self.manifest = file_prefix, file_suffix, manifest
        self.dtype = np.dtype(precision)
        self.staging_ring = bool(staging_ring)
        self._feeds = {}  # sample dtype -> HostFeed
        os.makedirs(out_dir, exist_ok=True)

    # -- host side ----------------------------------------------------------------------------

    def read(self, utt_id: str, path: str) -> np.ndarray:
        """One mono signal; channel rules and messages of the reference (command_line.py:104-126)"""
        try:
            # (native dtype: 16-bit PCM travels as int16 in float32 precision -- the cast the reference does right
            # here, command_line.py:107-108, happens on the device as the frames are loaded, exactly)
            signal = read_signal(path, dtype=None, force_as=self.force_as, key=utt_id)
        except Exception as exc:
            raise IOError(f"Utterance {utt_id}: {exc}") from exc
        if not (signal.dtype == np.int16 and self.dtype == np.float32):
            signal = signal.astype(self.dtype, copy=False)
        if self.channel == -1 and signal.ndim > 1 and signal.shape[0] > 1:
            raise ValueError(
                f"Utterance {utt_id}: Channel is not specified but signal has shape {signal.shape}")
        if (self.channel != -1 and signal.ndim == 1) or self.channel >= signal.shape[0]:
            raise ValueError(
                f"Utterance {utt_id}: Channel specified as {self.channel} but signal has shape {signal.shape}")
        if signal.ndim != 1:
            signal = signal[self.channel]
        return np.ascontiguousarray(signal)

    # -- device side --------------------------------------------------------------------------

    def process(self, signals: List[np.ndarray], first_index: int):
        """Features of one batch of host signals: a list of ``(T, F)`` float32 CPU tensors"""
        import torch

        lengths = np.asarray([len(s) for s in signals], dtype=np.int64)
        offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
        kinds = {s.dtype for s in signals}
        if len(kinds) > 1:  # (a batch of PCM and float files)
            signals = [s.astype(self.dtype, copy=False) for s in signals]
        through_feed = self._process_through_feed(signals, lengths)
        if through_feed is not None:
            return through_feed
        host = np.concatenate(signals) if offsets[-1] else np.zeros(0, self.dtype)
        packed = torch.from_numpy(host).to("cuda")
        if packed.dtype == torch.int16:
            packed = packed.to(torch.float32)  # (uploaded as PCM, widened on the device)
        fused = 0.0
        for k, pre in enumerate(self.pre):
            if isinstance(pre, Dither):
                for b in range(len(signals)):  # per-utterance stream, as the reference seeds torch
                    seg = packed[offsets[b] : offsets[b + 1]]
                    seg.copy_(Dither(pre.coeff, seed=self.seed + first_index + b).apply(seg))
            elif k == len(self.pre) - 1 and getattr(self.computer, "fuses_preemphasis", False):
                fused = pre.coeff  # rides along with the frame loads (STFT computers only)
            else:
                packed = pre.apply_packed(packed, offsets[:-1], lengths)
        if self.computer is None:
            feats, rows = packed.reshape(-1, 1), offsets
        else:
            kwargs = {"preemphasis": fused} if fused else {}
            feats, rows = self.computer.compute_packed(packed, offsets[:-1], lengths, **kwargs)
        feats, rows = self._postprocess(feats, np.asarray(rows, dtype=np.int64))
        feats = feats.to(torch.float32).cpu()
        return [feats[rows[b] : rows[b + 1]].clone() for b in range(len(signals))]

    def _process_through_feed(self, signals, lengths):
        """The batch through the pinned staging ring (``feed.HostFeed``) when the chain allows: an STFT computer
        with a fused kernel, float32 precision, no pre-processor but a pre-emphasis (which rides along with the frame
        loads); else ``None``"""
        import torch

        from .feed import HostFeed

        from . import config

        comp = self.computer
        if comp is None or self.dtype != np.float32 or not getattr(comp, "fuses_preemphasis", False) or not len(signals):
            return None
        if not (self.staging_ring and config.HOST_FEED):
            return None
        if len(self.pre) > 1 or (self.pre and not isinstance(self.pre[0], Preemphasize)) or not int(lengths.sum()):
            return None
        dtype = signals[0].dtype
        plan = comp._native_plan()
        if not plan.kernel_kind or (dtype == np.int16 and not plan.has_i16in) or dtype not in (np.float32, np.int16):
            return None
        C = comp.num_coeffs
        key = np.dtype(dtype)
        feed = self._feeds.get(key)
        need_samples, need_utts = int(lengths.sum()), len(signals)
        if feed is None or feed.slot_samples < need_samples or feed.slot_utts < need_utts:
            if feed is not None:
                feed.close()
            # room for what the post-processors make of a row (Deltas: (K + 1) C columns): tried on a few zero rows
            try:
                probe, _ = self._postprocess(torch.randn((64, C), device="cuda"), np.asarray([0, 64], dtype=np.int64))
                widen = -(-probe.numel() // (64 * C))
            except Exception:  # (a post-processor that wants more rows than the probe has)
                widen = 4
            feed = self._feeds[key] = HostFeed(comp, dtype, slot_samples=max(need_samples, 1 << 22), slot_utts=max(need_utts, 64),
                                               slots=2, copy_threads=min(8, os.cpu_count() or 1), feature_cols=widen * C + C)
        out_rows = {}

        def post(feats, row_offsets):
            feats, rows = self._postprocess(feats, np.asarray(row_offsets, dtype=np.int64))
            out_rows["rows"] = rows
            return feats

        # (no post-processor: the kernel writes the features into the pinned host buffer itself)
        ticket = feed.submit(signals, self.pre[0].coeff if self.pre else 0.0, post=post if self.post else None)
        feats, own_rows = feed.collect(ticket, copy=True)
        feats = torch.from_numpy(feats)
        rows = out_rows["rows"] if self.post else own_rows
        return [feats[rows[b] : rows[b + 1]].clone() for b in range(len(signals))]

    def _postprocess(self, feats, rows):
        import torch

        for post in self.post:
            if isinstance(post, Stack) and post.time_axis in (0, -2) and feats.dtype == torch.float32:
                try:
                    feats, rows = post.apply_rows(feats, rows)
                    continue
                except ValueError:
                    pass  # a pad mode the row kernel does not have: per utterance below
            elif isinstance(post, Standardize) and post.have_stats:
                feats = post.apply(feats, axis=-1)  # one global transform: the whole batch at once
                continue
            elif isinstance(post, Standardize) and feats.dtype == torch.float32 and feats.shape[0]:
                # (an utterance of fewer than two rows is an error of Standardize, raised below)
                if all(rows[b + 1] - rows[b] >= 2 for b in range(len(rows) - 1)):
                    feats = post.apply_rows(feats, rows)
                    continue
            # any other post-processor (Deltas acts along axis -1 here, exactly as the reference's
            # tool calls it): utterance by utterance on the device
            pieces = [post.apply(feats[rows[b] : rows[b + 1]]) for b in range(len(rows) - 1)]
            rows = np.concatenate([[0], np.cumsum([p.shape[0] for p in pieces])]).astype(np.int64)
            feats = torch.cat(pieces) if pieces else feats
        return feats, rows

    # -- the loop -------------------------------------------------------------------------------

    def run(self, utt_path: Sequence, batch_utts: int = 256, batch_samples: int = 1 << 26,
            num_workers: int = 0) -> int:
        """Process ``[(utt_id, path), ...]`` in order; returns the number of files written"""
        import torch

        pool = concurrent.futures.ThreadPoolExecutor(num_workers) if num_workers else None
        done, i, n = 0, 0, len(utt_path)
        pending, submitted = {}, 0  # reads in flight: at most two batches ahead of the GPU

        def top_up():
            nonlocal submitted
            while pool and submitted < n and submitted < i + 2 * batch_utts:
                pending[submitted] = pool.submit(self.read, *utt_path[submitted])
                submitted += 1

        writes = []  # (utt_id, future) of the batch before: files are written behind the GPU too

        def settle():
            nonlocal done
            for utt_id, fut in writes:
                fut.result()
                if self.manifest is not None:  # listed only once the file is complete
                    print(utt_id, file=self.manifest, flush=True)
                done += 1
            writes.clear()

        try:
            while i < n:
                ids, signals, total = [], [], 0
                first = i
                while i < n and len(ids) < batch_utts and (not ids or total < batch_samples):
                    utt_id, path = utt_path[i]
                    top_up()
                    signal = pending.pop(i).result() if pool else self.read(utt_id, path)
                    ids.append(utt_id)
                    signals.append(signal)
                    total += len(signal)
                    i += 1
                feats = self.process(signals, first)
                settle()
                for utt_id, feat in zip(ids, feats):
                    path = os.path.join(self.out_dir, self.prefix + utt_id + self.suffix)
                    if pool:
                        writes.append((utt_id, pool.submit(torch.save, feat, path)))
                    else:
                        torch.save(feat, path)
                        if self.manifest is not None:
                            print(utt_id, file=self.manifest, flush=True)
                        done += 1
            settle()
        finally:
            if pool:
                pool.shutdown(wait=True, cancel_futures=True)
        return done


def signals_to_torch_feat_dir(args=None) -> int:
    """Entry point; returns the process exit code (0 on success), like the reference's"""
    try:
        options = _parse(args)
    except SystemExit as ex:
        return ex.code
    seed = int(np.random.randint(np.iinfo(np.int32).max)) if options.seed is None else options.seed
    utt2path = {}
    for line_no, line in enumerate(options.map, 1):
        fields = line.strip().split(" ")
        if fields == [""]:
            continue
        if len(fields) < 2:
            print(f"Line {line_no} of {options.map.name}: not of format <utt_id> <path>", file=sys.stderr)
            return 1
        if fields[0] in utt2path:
            print(f'Line {line_no} of {options.map.name}: "{fields[0]}" already exists as utterance',
                  file=sys.stderr)
            return 1
        utt2path[fields[0]] = " ".join(fields[1:])
    if options.manifest is not None:
        options.manifest.seek(0)
        for line in options.manifest:
            utt2path.pop(line.strip(), None)
    computer = None
    if options.computer_config is not None:
        computer = alias_factory_subclass_from_arg(FrameComputer, options.computer_config)
    writer = FeatureDirWriter(
        computer, _as_list(options.preprocess, PreProcessor), _as_list(options.postprocess, PostProcessor),
        options.dir, options.channel, options.force_as, seed, options.file_prefix, options.file_suffix,
        options.manifest, options.precision, options.staging_ring,
    )
    writer.run(list(utt2path.items()), options.batch_utts, options.batch_samples, options.num_workers)
    return 0


if __name__ == "__main__":
    sys.exit(signals_to_torch_feat_dir())
