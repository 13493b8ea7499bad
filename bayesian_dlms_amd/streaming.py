"""Chain and data CSV I/O — the host-side mirror of the reference's `Streaming` helpers for the path's inputs and
outputs (Streaming.scala:25-81) and of the CSV layouts in examples/data (examples/.../FirstOrderDlm.scala:31-52).  Plain files,
no device work: the Gibbs drivers in `gibbs.py` yield DlmParameters, these functions put them on disk and back."""
import csv
import math
from typing import Callable, Iterable, Iterator, List, Optional, Sequence

import numpy as np

from .dlm import Data, DlmParameters


def write_chain(format_parameters: Callable[[DlmParameters], Sequence[float]], filename: str,
                iters: Iterable[DlmParameters], header: Optional[Sequence[str]] = None) -> int:
    """Streaming.writeChain (Streaming.scala:25-39): one CSV row per MCMC iteration, written as the iterator is
    consumed (the chain is never held in memory).  `header` plays the role of the CsvConfiguration's header.
    Returns the number of rows written."""
    n = 0
    with open(filename, "w", newline="") as fh:
        w = csv.writer(fh)
        if header is not None:
            w.writerow(list(header))
        for p in iters:
            w.writerow([repr(float(x)) for x in format_parameters(p)])
            n += 1
    return n


def read_mcmc_chain(filename: str, header: bool = True) -> Iterator[List[float]]:
    """Streaming.readMcmcChain (Streaming.scala:62-65): rows of doubles, the header skipped (rfc.withHeader)."""
    with open(filename, newline="") as fh:
        r = csv.reader(fh)
        if header:
            next(r, None)
        for row in r:
            if row:
                yield [float(x) for x in row]


def parse_diagonal_parameters(v_dim: int, w_dim: int, ps: Sequence[float]) -> DlmParameters:
    """Streaming.parseDiagonalParameters (Streaming.scala:45-57): diag(v), diag(w), m0, then the FULL c0 in
    column-major order (Breeze's `new DenseMatrix(wDim, wDim, c0)`), unlike DlmParameters.fromList's diagonal c0."""
    ps = np.asarray(ps, dtype=np.float64)
    need = v_dim + 2 * w_dim + w_dim * w_dim
    if ps.size < need:
        raise ValueError(f"expected at least {need} values, got {ps.size}")
    v = ps[:v_dim]
    w = ps[v_dim:v_dim + w_dim]
    m0 = ps[v_dim + w_dim:v_dim + 2 * w_dim]
    c0 = ps[v_dim + 2 * w_dim:need].reshape(w_dim, w_dim).T
    return DlmParameters(np.diag(v), np.diag(w), m0, c0)


def col_means(params: Sequence[Sequence[float]]) -> List[float]:
    """Streaming.colMeans (Streaming.scala:70-72)."""
    return [float(x) for x in np.mean(np.asarray(params, dtype=np.float64), axis=0)]


def quantile(xs: Sequence[float], prob: float):
    """Streaming.quantile (Streaming.scala:74-78): the element at floor(n * prob) of the sorted sample (no
    interpolation; prob = 1 is out of range, as in the reference)."""
    ordered = sorted(xs)
    return ordered[int(math.floor(len(ordered) * prob))]


def mean_parameters(iters: Iterable[DlmParameters], v_dim: int, w_dim: int) -> DlmParameters:
    """Streaming.meanParameters (Streaming.scala:111-118), literally: the fold starts from the EMPTY parameter set
    with count 1, so the result is sum / (n + 1), not the arithmetic mean (the reference's off-by-one)."""
    avg = DlmParameters(np.zeros((v_dim, v_dim)), np.zeros((w_dim, w_dim)), np.zeros(w_dim), np.zeros((w_dim, w_dim)))
    n = 1.0
    for b in iters:
        avg = DlmParameters((avg.v * n + b.v) / (n + 1), (avg.w * n + b.w) / (n + 1),
                            (avg.m0 * n + b.m0) / (n + 1), (avg.c0 * n + b.c0) / (n + 1))
        n += 1
    return avg


def write_simulated(filename: str, times: Sequence[float], x: np.ndarray, y: np.ndarray) -> None:
    """The examples' simulated-data layout (SimulateDlm, FirstOrderDlm.scala:39-50; examples/data/first_order_dlm.csv):
    header `time,observation[_j],state[_i]`, one row per observation time; a missing observation is written as NaN
    (KalmanFilter.flattenObs)."""
    y = np.asarray(y, dtype=np.float64).reshape(len(times), -1)
    x = np.asarray(x, dtype=np.float64).reshape(len(times), -1)
    oh = ["observation"] if y.shape[1] == 1 else [f"observation_{j + 1}" for j in range(y.shape[1])]
    sh = ["state"] if x.shape[1] == 1 else [f"state_{i + 1}" for i in range(x.shape[1])]
    with open(filename, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["time"] + oh + sh)
        for t, yr, xr in zip(times, y, x):
            w.writerow([repr(float(t))] + ["NaN" if math.isnan(v) else repr(float(v)) for v in yr] + [repr(float(v)) for v in xr])


def read_data(filename: str, n_obs: int = 1) -> List[Data]:
    """Read `time, observation...` rows back as Data (SimulatedData, FirstOrderDlm.scala:31-37: columns 0 and 1 for a
    univariate model); unparsable cells (NA, empty) become missing observations."""
    out = []
    with open(filename, newline="") as fh:
        r = csv.reader(fh)
        next(r, None)
        for row in r:
            if not row:
                continue
            obs = []
            for cell in row[1:1 + n_obs]:
                try:
                    obs.append(float(cell))
                except ValueError:
                    obs.append(float("nan"))
            out.append(Data(float(row[0]), np.asarray(obs, dtype=np.float64)))
    return out
