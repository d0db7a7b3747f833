"""``--profile`` of the stage-1 trainer on the launch plans' own per-op timeline.

The reference wraps a few steps in ``torch.profiler`` (3d_ldm/train_autoencoder.py:312-329: ``schedule(wait=1, warmup=1, active=3,
repeat=2)``, traces to ``./profiler_logs``, ``profiler.step()`` once per training step :450).  Here a training step is two launch
plans of the library (forward, backward) plus the optimizer launches, so the equivalent record is the library's per-op timeline
(``ldm_set_plan_trace``: a HIP event in front of every op of every plan that runs while it is on).  ``PlanProfiler`` keeps the
reference's schedule: per cycle 1 step ignored, 1 warm-up step, 3 traced steps, 2 cycles; one CSV per cycle under ``out_dir`` and a
per-kind summary on rank 0.
"""
from __future__ import annotations

import collections
import os
from typing import Dict, List, Optional

from . import _lib

# OpKind of csrc/ldm3d.hip, in enum order (the third CSV column)
KINDS = ("PACK CONV FINALIZE GN_STATS GN_FINALIZE GN_PREP GN_APPLY ATTN SINUSOID GEMV VAE_HEADS GN_FUSED WT WT_BATCH WGRAD EXPORT "
         "EXPORT_BATCH COLSUM GNB ATTN_BWD ADD SUMPOOL LIN_DX LIN_DW VAE_HEADS_BWD GEMM_LIGHT COLSUM_BATCH IM2COL "
         "PACK32 CONV32 FIN32 GN_STATS32 GN_APPLY32 ATTN32 GEMV32 TAP BUCKET BUCKET_JOIN UPS_SPLIT32 CONV_THIN FIN_GN GEMM_LIGHT32 CONV_BLOCK TEMB_ROW").split()


def summarize(path: str) -> Dict[str, List[float]]:
    """{op kind: [launch-plan ops, total microseconds]} of one trace CSV (rows: ops in plan, op index, kind, us, description)."""
    out: Dict[str, List[float]] = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as fh:
        for ln in fh:
            parts = ln.rstrip("\n").split(",", 4)
            if len(parts) < 4:
                continue
            k = int(parts[2])
            name = KINDS[k] if 0 <= k < len(KINDS) else f"KIND{k}"
            out[name][0] += 1
            out[name][1] += float(parts[3])
    return dict(out)


class PlanProfiler:
    def __init__(self, out_dir: str = "./profiler_logs", wait: int = 1, warmup: int = 1, active: int = 3, repeat: int = 2,
                 verbose: bool = True):
        self.out_dir, self.wait, self.warmup, self.active, self.repeat, self.verbose = out_dir, wait, warmup, active, repeat, verbose
        os.makedirs(out_dir, exist_ok=True)
        self.step_idx = 0
        self.paths: List[str] = []
        self._current: Optional[str] = None
        self._apply()

    def _phase(self):
        cycle_len = self.wait + self.warmup + self.active
        cycle, pos = divmod(self.step_idx, cycle_len)
        if cycle >= self.repeat:
            return None
        return cycle if pos >= self.wait + self.warmup else None

    def _apply(self):
        cycle = self._phase()
        want = None if cycle is None else os.path.join(self.out_dir, f"plan_trace_cycle{cycle}.csv")
        if want == self._current:
            return
        L = _lib.lib()
        if self._current is not None and want != self._current:
            _lib.check(L.ldm_set_plan_trace(None))
            self._report(self._current)
        if want is not None:
            if os.path.exists(want):
                os.remove(want)
            _lib.check(L.ldm_set_plan_trace(want.encode()))
            self.paths.append(want)
        self._current = want

    def _report(self, path: str):
        if not self.verbose or not os.path.exists(path):
            return
        groups = summarize(path)
        total = sum(v[1] for v in groups.values()) or 1.0
        print(f"[profile] {path}: {total / max(1, self.active) / 1e3:.3f} ms of launch-plan ops per traced step")
        for name, (n, us) in sorted(groups.items(), key=lambda kv: -kv[1][1])[:12]:
            print(f"[profile]   {name:14s} {us / self.active:10.1f} us/step  {100 * us / total:5.1f} %  ({n // max(1, self.active)} ops/step)")

    def step(self):
        """Call once per training step (the reference's ``profiler.step()``)."""
        self.step_idx += 1
        self._apply()

    def stop(self):
        if self._current is not None:
            _lib.check(_lib.lib().ldm_set_plan_trace(None))
            self._report(self._current)
            self._current = None
