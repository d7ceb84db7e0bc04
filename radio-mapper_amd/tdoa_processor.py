"""Host-side mirror of the reference's ``tdoa_processor.py`` interface, with the pair loop's time
difference measured from IQ on the GPU.

Same public names, argument meaning and error behaviour as the reference module so that
``central_processor.py`` / the reference's tests can switch imports:

  reference item (tdoa_processor.py)                      here
  -----------------------------------------------------  ------------------------------------------
  BuoyPosition / SignalDetection / TDoAMeasurement /      same field names and order (:24-69);
  TriangulationResult dataclasses                          SignalDetection gains two OPTIONAL trailing
                                                           fields (iq_samples, sample_rate_hz)
  GeodeticCalculator (:71-136)                             same statics (spherical ECEF, R=6378137)
  TDoACalculator.calculate_tdoa_measurements (:146-198)    same pair order / skips / confidence;
                                                           ``time_diff_ns`` (:166) becomes
                                                           window-start difference + xcorr lag when
                                                           every detection carries IQ
  HyperbolicPositioning.triangulate_position (:218-328)    same objective, BFGS, result fields
  TDoAProcessor (:330-465)                                 same ctor / register_buoy /
                                                           process_signal_detections / status

Without IQ on the detections every number equals the reference's (pinned by
tests/golden/tdoa_conventions.json).  With IQ the lag comes from ``xcorr.XcorrEngine`` (HIP, gfx950);
there is no CPU fallback and no silent degradation to timestamps: if IQ is supplied and the HIP library
or a GPU is missing -- or only some detections of a group carry IQ, or their windows differ in length, dtype or
sample rate -- the seam logs the error on ``...TDoACalculator`` and yields no measurements for that group (the reference's log-and-return convention, tdoa_processor.py:151-153);
``TDoACalculator.measure_lags`` itself raises (ImportError / RmxError).
"""
from __future__ import annotations

import logging
import math
from dataclasses import dataclass
from datetime import datetime, timezone
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

logger = logging.getLogger(__name__)

_C = 299792458.0


# --------------------------------------------------------------------------------------------------
# data model (field names/order: tdoa_processor.py:24-69; tests construct these positionally)
# --------------------------------------------------------------------------------------------------
@dataclass
class BuoyPosition:
    buoy_id: str
    lat: float
    lng: float
    altitude: float = 0.0
    timing_accuracy_ns: int = 100000


@dataclass
class SignalDetection:
    buoy_id: str
    frequency_mhz: float
    signal_strength_dbm: float
    timestamp_utc: str
    gps_timestamp_ns: int
    lat: float
    lng: float
    confidence: float
    signal_type: str = "unknown"
    # extension (not in the reference): the capture window this detection was made on.
    # gps_timestamp_ns is then the time tag of the window's first sample.
    iq_samples: Optional[Any] = None          # np.ndarray complex64 [N] or raw uint8 [2N]
    sample_rate_hz: Optional[float] = None


@dataclass
class TDoAMeasurement:
    buoy1_id: str
    buoy2_id: str
    time_difference_ns: int        # buoy2 - buoy1; > 0 when buoy2 received later
    distance_difference_m: float
    confidence: float
    frequency_mhz: float


@dataclass
class TriangulationResult:
    estimated_lat: float
    estimated_lng: float
    estimated_altitude: float
    accuracy_meters: float
    confidence: float
    frequency_mhz: float
    signal_type: str
    timestamp_utc: str
    contributing_buoys: List[str]
    tdoa_measurements: List[TDoAMeasurement]
    method: str

    @property
    def accuracy_estimate_meters(self) -> float:
        """Name central_processor.py:434 reads (absent in the reference: SURVEY.md section 0.2)."""
        return self.accuracy_meters


class GeodeticCalculator:
    """Spherical-earth helpers with the reference's constants (tdoa_processor.py:71-136)."""

    EARTH_RADIUS_M = 6378137.0

    @staticmethod
    def lat_lng_to_xyz(lat: float, lng: float, alt: float = 0.0) -> Tuple[float, float, float]:
        phi, lam = math.radians(lat), math.radians(lng)
        r = GeodeticCalculator.EARTH_RADIUS_M + alt
        c = math.cos(phi)
        return r * c * math.cos(lam), r * c * math.sin(lam), r * math.sin(phi)

    @staticmethod
    def xyz_to_lat_lng(x: float, y: float, z: float) -> Tuple[float, float, float]:
        rho = math.sqrt(x * x + y * y)
        return (math.degrees(math.atan2(z, rho)), math.degrees(math.atan2(y, x)),
                math.sqrt(x * x + y * y + z * z) - GeodeticCalculator.EARTH_RADIUS_M)

    @staticmethod
    def distance_3d(lat1, lng1, alt1, lat2, lng2, alt2) -> float:
        a = GeodeticCalculator.lat_lng_to_xyz(lat1, lng1, alt1)
        b = GeodeticCalculator.lat_lng_to_xyz(lat2, lng2, alt2)
        return math.sqrt(sum((q - p) ** 2 for p, q in zip(a, b)))

    @staticmethod
    def bearing_distance(lat1, lng1, lat2, lng2) -> Tuple[float, float]:
        p1, p2 = math.radians(lat1), math.radians(lat2)
        dl = math.radians(lng2 - lng1)
        h = math.sin((p2 - p1) / 2) ** 2 + math.cos(p1) * math.cos(p2) * math.sin(dl / 2) ** 2
        dist = GeodeticCalculator.EARTH_RADIUS_M * 2 * math.atan2(math.sqrt(h), math.sqrt(1 - h))
        brg = math.atan2(math.sin(dl) * math.cos(p2),
                         math.cos(p1) * math.sin(p2) - math.sin(p1) * math.cos(p2) * math.cos(dl))
        return (math.degrees(brg) + 360) % 360, dist


# --------------------------------------------------------------------------------------------------
# the seam: pairwise TDoA
# --------------------------------------------------------------------------------------------------
class TDoACalculator:
    SPEED_OF_LIGHT = _C  # tdoa_processor.py:141
    MAX_ENGINES = 4      # engines kept alive (one rmx_ctx each: device scratch), least recently used evicted

    def __init__(self, device: int = 0, devices: Optional[Sequence[int]] = None, min_cut_samples: int = 128):
        """device: the GPU of a single-device calculator (the default).  devices: a list of GPUs, or "all" for every
        visible one -- with more than one entry a batch of windows / frequency groups is block-sharded over them by
        `multi.MultiXcorrEngine` (one rmx_ctx and one host thread per device, no collective).  min_cut_samples: the
        shortest window a group may be CUT to when its windows differ in length (see _group_windows); 128 is the
        shortest cut the reference's own clipping can produce."""
        self.logger = logging.getLogger(__name__ + ".TDoACalculator")
        self.device = device
        self.devices = devices
        self.min_cut_samples = int(min_cut_samples)
        self._engines: Dict[Tuple[int, int], Any] = {}   # insertion order = recency
        self._tconf: Dict[Tuple[int, int], float] = {}

    # -- GPU engine cache ------------------------------------------------------------------------
    def _engine(self, n_buoys: int, n_samples: int, n_windows: int = 1):
        from . import xcorr  # raises ImportError loudly if the HIP library is not built
        key = (n_buoys, n_samples)
        eng = self._engines.pop(key, None)
        if eng is None or eng.max_windows < n_windows:
            if eng is not None:
                eng.close()
            # head room: the number of frequency groups varies from call to call at the seam, and an engine is only
            # rebuilt (milliseconds: tables, scratch) when a batch exceeds every earlier one by a power of two
            cap = 8
            while cap < n_windows:
                cap *= 2
            n_windows = cap
            devs = self.devices
            if isinstance(devs, str):
                if devs != "all":
                    raise ValueError('devices must be a list of device indices or "all"')
                devs = list(range(xcorr.device_count()))
            if devs is not None and len(devs) > 1:
                from . import multi
                eng = multi.MultiXcorrEngine(n_buoys, n_samples, max(n_windows, 1), devices=devs)
            else:
                eng = xcorr.XcorrEngine(n_buoys, n_samples, max(n_windows, 1),
                                        device=self.device if not devs else int(devs[0]))
        self._engines[key] = eng                         # most recently used last
        while len(self._engines) > self.MAX_ENGINES:
            old = next(iter(self._engines))
            self._engines.pop(old).close()
        return eng

    def close(self):
        for eng in self._engines.values():
            eng.close()
        self._engines.clear()

    def measure_lags(self, iq, pairs=None):
        """Batched hot path.  iq: complex64 [W][B][N] (or uint8 [W][B][2N]) ->
        (lag_int [W][P], lag_frac [W][P], peak [W][P]); lag = delay(j) - delay(i) in samples.
        A leading channel axis is a batch axis: [C][W][B][N] -> three [C][W][P] arrays (channels and
        windows are independent units, tdoa_processor.py:363)."""
        iq = np.asarray(iq)
        lead = None
        if iq.ndim == 4:
            lead = iq.shape[:2]
            iq = iq.reshape((lead[0] * lead[1],) + iq.shape[2:])
        if iq.ndim != 3:
            raise ValueError(f"iq must be [W][B][N] or [C][W][B][N], got shape {iq.shape}")
        n = iq.shape[2] // 2 if iq.dtype == np.uint8 else iq.shape[2]
        out = self._engine(iq.shape[1], n, iq.shape[0]).correlate(iq, pairs)
        if lead is not None:
            out = tuple(a.reshape(lead + a.shape[1:]) for a in out)
        return out

    @staticmethod
    def _iq_of(det: SignalDetection):
        s = det.iq_samples
        if s is None or det.sample_rate_hz is None:
            return None
        if isinstance(s, (list, tuple)) and len(s) and isinstance(s[0], (str, list, tuple)):
            # the reference's wire form: NumpyEncoder turns every complex into str(obj) (iq_stream_client.py:31-44)
            from .iq_wire import parse_complex_list
            return parse_complex_list(s)
        a = np.asarray(s)
        if a.dtype == np.uint8:
            return a
        return np.ascontiguousarray(a, dtype=np.complex64)

    def _iq_batch_key(self, detections: Sequence[SignalDetection]):
        """(key, stacked [B][N] windows) when every detection of the group carries an IQ window of one
        shape, dtype and sample rate; (None, None) when none does; (False, None) when they disagree."""
        try:
            iqs = [self._iq_of(d) for d in detections]
        except ValueError as e:      # wire data that does not parse (iq_wire.IqWireError): logged, no measurements
            self.logger.error(f"IQ samples of the group cannot be decoded: {e}")
            return False, None
        if all(a is None for a in iqs):
            return None, None
        # IQ was supplied for this group: from here on the time tags alone are never used for it.  A group that
        # cannot be correlated as one [B][N] batch yields no measurements (logged), as an engine failure does.
        if any(a is None for a in iqs):
            self.logger.error("Only some detections of the group carry IQ windows; no TDoA measurements for it")
            return False, None
        rates = {float(d.sample_rate_hz) for d in detections}
        kinds = {(a.dtype.str, a.ndim) for a in iqs}
        if len(kinds) != 1 or len(rates) != 1 or iqs[0].ndim != 1:
            self.logger.error("IQ windows of the group differ in dtype/sample rate; no TDoA measurements for it")
            return False, None
        # Windows that differ in length, or whose length is not a power of two: the reference clips an excerpt at the
        # end of the capture buffer (iq_stream_client.py:306-313: start = max(0, peak - 128), end = min(len, start + 256)),
        # so a peak in the last 128 bins gives e.g. 130 samples beside the other buoys' 256.  Every window keeps its
        # START -- under this build's extension the time tag of a detection that carries IQ dates its window's first
        # sample (SignalDetection above; the reference's own tag is time.time_ns() at detection time,
        # iq_stream_client.py:221, and dates no sample) -- so the group is cut to the largest power of two that every
        # window holds -- the engine's window lengths -- instead of being dropped.  The reference's clipping
        # never leaves fewer than 129 samples, so a cut below min_cut_samples (128) is not one of its excerpts: such a
        # group is refused rather than correlated on a noise-dominated stub.  Windows of one equal
        # power-of-two length >= 16 are the caller's own choice and are not cut.
        per = 2 if iqs[0].dtype == np.uint8 else 1          # uint8: interleaved I, Q
        nmin = min(a.shape[0] // per for a in iqs)
        n = 1 << (max(nmin, 1).bit_length() - 1)
        if nmin < 16:
            self.logger.error(f"IQ windows of the group are too short to correlate ({nmin} samples); no TDoA measurements for it")
            return False, None
        if any(a.shape[0] != n * per for a in iqs):
            if n < self.min_cut_samples:
                self.logger.error(f"IQ windows of the group hold {sorted({a.shape[0] // per for a in iqs})} samples: their "
                                  f"common cut of {n} is below {self.min_cut_samples}; no TDoA measurements for it")
                return False, None
            self.logger.warning(f"IQ windows of the group hold {sorted({a.shape[0] // per for a in iqs})} samples; "
                                f"correlating their first {n}")
            iqs = [a[:n * per] for a in iqs]
        return (len(iqs), iqs[0].dtype.str, iqs[0].shape, next(iter(rates))), np.stack(iqs)

    def _measure_groups(self, stacked: np.ndarray):
        """[G][B][N] -> lag [G][P] float64, or None after logging: the reference's seam never raises
        (tdoa_processor.py:151-153) and there is no fallback to time tags once IQ was supplied."""
        try:
            li, lf, _ = self.measure_lags(stacked)
            return li.astype(np.float64) + lf.astype(np.float64)
        except (ImportError, OSError) as e:   # library not built / not loadable
            self.logger.error(f"Cross-correlation engine failed: {e}")
            return None
        except Exception as e:
            from . import xcorr
            if not isinstance(e, xcorr.RmxError):    # programming errors (shapes, types) surface
                raise
            self.logger.error(f"Cross-correlation engine failed: {e}")   # no device, engine refused the batch
            return None

    def _timing_confidence(self, b1: BuoyPosition, b2: BuoyPosition) -> float:
        # exp(-rss(timing accuracies) / 100 us), capped at 1 (tdoa_processor.py:200-210); a handful of distinct values
        key = (b1.timing_accuracy_ns, b2.timing_accuracy_ns)
        c = self._tconf.get(key)
        if c is None:
            if len(self._tconf) > 4096:
                self._tconf.clear()
            c = self._tconf[key] = min(math.exp(-math.hypot(key[0], key[1]) / 100000), 1.0)
        return c

    _calculate_timing_confidence = _timing_confidence  # reference's private name

    _NO_LAG = object()      # calculate_tdoa_measurements inspects the detections itself
    _TIME_TAGS = object()   # the caller already did and found no IQ: the reference's arithmetic (:166)

    def calculate_tdoa_measurements(self, detections: List[SignalDetection],
                                    buoy_positions: Dict[str, BuoyPosition],
                                    _lag=_NO_LAG) -> List[TDoAMeasurement]:
        """The reference's pair loop (tdoa_processor.py:146-198).  `_lag` (private): (lags [P], sample rate) of this
        group, already measured in a batch with other groups by TDoAProcessor; None when that batch failed or the
        group's IQ could not be batched; `_TIME_TAGS` when TDoAProcessor found no IQ on the group."""
        out: List[TDoAMeasurement] = []
        nd = len(detections)
        if nd < 2:
            self.logger.warning("Need at least 2 detections for TDoA calculation")
            return out
        lag = None
        fs = None
        if _lag is self._NO_LAG:
            key, stacked = self._iq_batch_key(detections)
            if key is False:
                return out
            if key:
                res = self._measure_groups(stacked[None])
                if res is None:
                    return out
                lag, fs = res[0], key[3]
        elif _lag is None:
            return out
        elif _lag is not self._TIME_TAGS:
            lag, fs = _lag
        q = -1
        if lag is not None:
            # the same float64 arithmetic as `round(lag[q] / fs * 1e9)` per pair, done once for the group (numpy scalars make
            # the pair loop four times slower than it needs to be; np.rint and round() both round half to even)
            lag_s = np.asarray(lag, np.float64) / fs
            if not np.all(np.isfinite(lag_s)):           # (a NaN would become INT64_MIN in the cast below)
                self.logger.error("Cross-correlation returned a non-finite lag; no TDoA measurements for the group")
                return out
            lag_ns = np.rint(lag_s * 1e9).astype(np.int64).tolist()
        debug = self.logger.isEnabledFor(logging.DEBUG)
        for i in range(nd):
            for j in range(i + 1, nd):
                q += 1
                d1, d2 = detections[i], detections[j]
                if abs(d1.frequency_mhz - d2.frequency_mhz) > 0.01:
                    continue
                dt_ns = d2.gps_timestamp_ns - d1.gps_timestamp_ns
                if lag is not None:
                    dt_ns += lag_ns[q]
                dist_m = dt_ns / 1e9 * self.SPEED_OF_LIGHT
                p1, p2 = buoy_positions.get(d1.buoy_id), buoy_positions.get(d2.buoy_id)
                if not p1 or not p2:
                    continue
                conf = min(d1.confidence, d2.confidence) * self._timing_confidence(p1, p2)
                out.append(TDoAMeasurement(d1.buoy_id, d2.buoy_id, dt_ns, dist_m, conf, d1.frequency_mhz))
                if debug:
                    self.logger.debug("TDoA %s-%s dT=%.1f us dD=%.1f m", d1.buoy_id, d2.buoy_id, dt_ns / 1000, dist_m)
        return out


# --------------------------------------------------------------------------------------------------
# consumer: hyperbolic fix (unchanged semantics; tdoa_processor.py:218-328)
# --------------------------------------------------------------------------------------------------
class HyperbolicPositioning:
    def __init__(self):
        self.logger = logging.getLogger(__name__ + ".HyperbolicPositioning")

    def triangulate_position(self, measurements: List[TDoAMeasurement],
                             buoy_positions: Dict[str, BuoyPosition]) -> Optional[TriangulationResult]:
        if len(measurements) < 2:
            self.logger.warning("Need at least 2 TDoA measurements for triangulation")
            return None
        ids = set()
        for m in measurements:
            ids.update((m.buoy1_id, m.buoy2_id))
        if len(ids) < 3:
            self.logger.warning("Need at least 3 buoys for 2D triangulation")
            return None
        xyz = {}
        for b in ids:
            if b not in buoy_positions:
                self.logger.error(f"Missing position for buoy {b}")
                return None
            pos = buoy_positions[b]
            xyz[b] = GeodeticCalculator.lat_lng_to_xyz(pos.lat, pos.lng, pos.altitude)
        # the objective in the reference's own scalar arithmetic (math.sqrt, one division per term,
        # left-to-right sum: tdoa_processor.py:249-273): BFGS with finite-difference gradients stops on
        # "precision loss" or not depending on the last bits of this function, so a vectorised sum
        # (different rounding order) changes which calls return None
        terms = [(xyz[m.buoy1_id], xyz[m.buoy2_id], m.distance_difference_m, m.confidence + 0.1)
                 for m in measurements]

        def cost(tx):
            x, y, z = tx
            total = 0
            for a, b, meas_m, den in terms:
                d1 = math.sqrt((x - a[0]) ** 2 + (y - a[1]) ** 2 + (z - a[2]) ** 2)
                d2 = math.sqrt((x - b[0]) ** 2 + (y - b[1]) ** 2 + (z - b[2]) ** 2)
                total = total + ((d2 - d1) - meas_m) ** 2 / den
            return total

        pts = list(xyz.values())
        x0 = [sum(q[k] for q in pts) / len(pts) for k in range(3)]
        try:
            import scipy.optimize
            res = scipy.optimize.minimize(cost, x0, method="BFGS", options={"maxiter": 1000})
            if not res.success:
                self.logger.warning(f"Optimization failed: {res.message}")
                return None
            lat, lng, alt = GeodeticCalculator.xyz_to_lat_lng(*res.x)
            acc = math.sqrt(res.fun / len(measurements))
            conf = sum(m.confidence for m in measurements) / len(measurements)
            self.logger.info(f"Triangulation successful: ({lat:.6f}, {lng:.6f}) ±{acc:.1f}m, confidence: {conf:.2f}")
            return TriangulationResult(lat, lng, alt, acc, conf, measurements[0].frequency_mhz, "unknown",
                                       datetime.now(timezone.utc).isoformat(), list(ids), measurements,
                                       "hyperbolic")
        except Exception as e:  # reference convention: log, never raise (:326-328)
            self.logger.error(f"Triangulation failed: {e}")
            return None


    def triangulate_batch(self, engine, buoys: List[BuoyPosition], lag_int, lag_frac, sample_rate_hz: float,
                          confidence=None, max_iter: int = 60):
        """Batched form of triangulate_position on the GPU (rmx_solve_batch): one solve per window
        from the lag arrays XcorrEngine.correlate returned for `buoys` (detection-list order, all
        pairs i<j).  Same objective, start point and accuracy formula as the reference
        (tdoa_processor.py:249-300).  Returns a list of (lat, lng, altitude, accuracy_meters)."""
        xyz = np.array([GeodeticCalculator.lat_lng_to_xyz(b.lat, b.lng, b.altitude) for b in buoys])
        weight = None if confidence is None else 1.0 / (np.asarray(confidence, np.float64) + 0.1)
        pos, cost, _ = engine.solve(xyz, lag_int, lag_frac, sample_rate_hz, weight=weight, max_iter=max_iter)
        n_meas = np.asarray(lag_int).shape[1]
        return [GeodeticCalculator.xyz_to_lat_lng(*p) + (math.sqrt(max(c, 0.0) / n_meas),)
                for p, c in zip(pos, cost)]


# --------------------------------------------------------------------------------------------------
# orchestrator (tdoa_processor.py:330-465)
# --------------------------------------------------------------------------------------------------
class TDoAProcessor:
    def __init__(self):
        self.logger = logging.getLogger(__name__ + ".TDoAProcessor")
        self.tdoa_calculator = TDoACalculator()
        self.hyperbolic_positioner = HyperbolicPositioning()
        self.buoy_positions: Dict[str, BuoyPosition] = {}
        self.correlation_window_s = 10.0
        self.min_buoys_for_triangulation = 3

    def register_buoy(self, buoy_position: BuoyPosition):
        self.buoy_positions[buoy_position.buoy_id] = buoy_position
        self.logger.info(f"Registered buoy {buoy_position.buoy_id} at "
                         f"({buoy_position.lat:.6f}, {buoy_position.lng:.6f})")

    def _group_by_frequency(self, detections: Sequence[SignalDetection],
                            frequency_tolerance_mhz: float = 0.01) -> Dict[float, List[SignalDetection]]:
        """First-fit grouping on the group's first frequency (tdoa_processor.py:405-425)."""
        groups: Dict[float, List[SignalDetection]] = {}
        for d in detections:
            key = next((f for f in groups if abs(d.frequency_mhz - f) <= frequency_tolerance_mhz), None)
            if key is None:
                groups[d.frequency_mhz] = [d]
            else:
                groups[key].append(d)
        return groups

    def _filter_by_time_window(self, detections: Sequence[SignalDetection]) -> List[SignalDetection]:
        """Keep detections no older than correlation_window_s before the newest, sorted by time
        (tdoa_processor.py:427-445)."""
        if not detections:
            return []
        ordered = sorted(detections, key=lambda d: d.gps_timestamp_ns)
        cutoff = ordered[-1].gps_timestamp_ns - int(self.correlation_window_s * 1e9)
        return [d for d in ordered if d.gps_timestamp_ns >= cutoff]

    def process_signal_detections(self, detections: List[SignalDetection]) -> List[TriangulationResult]:
        if not detections:
            return []
        self.logger.info(f"Processing {len(detections)} signal detections")
        # frequency groups are independent units (tdoa_processor.py:363-377): all groups whose
        # detections carry IQ windows of one shape go to the GPU as ONE [groups][B][N] batch
        work = []
        for freq, group in self._group_by_frequency(detections).items():
            recent = self._filter_by_time_window(group)
            if len(recent) < self.min_buoys_for_triangulation:
                self.logger.debug(f"Insufficient detections for {freq} MHz ({len(recent)} < "
                                  f"{self.min_buoys_for_triangulation})")
                continue
            key, stacked = self.tdoa_calculator._iq_batch_key(recent)   # inspected (and logged) once per group
            work.append([freq, recent, key, stacked, TDoACalculator._TIME_TAGS if key is None else None])
        batches: Dict[Any, List[int]] = {}
        for n, item in enumerate(work):
            if item[2]:
                batches.setdefault(item[2], []).append(n)
        for key, members in batches.items():
            lags = self.tdoa_calculator._measure_groups(np.stack([work[n][3] for n in members]))
            for k, n in enumerate(members):
                work[n][4] = None if lags is None else (lags[k], key[3])
        results: List[TriangulationResult] = []
        for freq, recent, key, _, lag in work:
            meas = self.tdoa_calculator.calculate_tdoa_measurements(recent, self.buoy_positions, _lag=lag)
            if len(meas) < 2:
                self.logger.debug(f"Insufficient TDoA measurements for {freq} MHz")
                continue
            fix = self.hyperbolic_positioner.triangulate_position(meas, self.buoy_positions)
            if fix:
                kinds = [d.signal_type for d in recent]
                fix.signal_type = max(set(kinds), key=kinds.count)
                results.append(fix)
                if fix.signal_type == "emergency":
                    self.logger.warning(f"EMERGENCY SIGNAL TRIANGULATED: {freq} MHz at "
                                        f"({fix.estimated_lat:.6f}, {fix.estimated_lng:.6f}) "
                                        f"±{fix.accuracy_meters:.1f}m")
        return results

    def triangulate_signal(self, detections: List[SignalDetection]) -> Optional[TriangulationResult]:
        """What central_processor.py:418 calls (missing in the reference): first fix or None."""
        res = self.process_signal_detections(detections)
        return res[0] if res else None

    def get_buoy_network_status(self) -> Dict:
        return {
            "registered_buoys": len(self.buoy_positions),
            "buoy_list": [{"buoy_id": p.buoy_id, "lat": p.lat, "lng": p.lng,
                           "timing_accuracy_ns": p.timing_accuracy_ns}
                          for p in self.buoy_positions.values()],
            "min_buoys_required": self.min_buoys_for_triangulation,
            "correlation_window_s": self.correlation_window_s,
            "triangulation_ready": len(self.buoy_positions) >= self.min_buoys_for_triangulation,
        }
