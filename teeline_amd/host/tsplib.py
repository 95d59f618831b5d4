"""TSPLIB reader — mirror of src/tsp/tsplib.rs (state machine :142-255, weight formats :262-320).

Observable behaviour kept: lines trimmed and upper-cased; a bare identifier line starts a section;
coordinates parsed as f32; EXPLICIT weights (FULL_MATRIX / UPPER_ROW / LOWER_DIAG_ROW) repacked to the
strict lower triangle; unknown EDGE_WEIGHT_TYPE (e.g. ATT) silently becomes EUC_2D (:199-202); ATSP
rejected (:195-197); placeholder grid coordinates when only weights are present (:246-258).
"""
import re

import numpy as np

_SECTION = re.compile(r"^(?P<key>[A-Z_]\w*)$")
_KV = re.compile(r"^(?P<key>\w+)\s*:\s*(?P<val>.+)$")


class TspLibData:
    def __init__(self, name, comment, ids, xy, dimension, raw_distances, distance_type):
        self.name, self.comment = name, comment
        self.ids = np.asarray(ids, dtype=np.int64)
        self.xy = np.ascontiguousarray(np.asarray(xy, dtype=np.float32).reshape(-1, 2))
        self.dimension = dimension
        self.raw_distances = raw_distances
        self.distance_type = distance_type  # "euc2d" | "geo" | "explicit"

    def __len__(self):
        return len(self.ids)

    def has_explicit_weights(self):
        return self.raw_distances is not None

    def distance_matrix(self, ctx=None):
        """TspLibData::distance_matrix (tsplib.rs:84-98)."""
        from . import distance_matrix as dm
        if self.raw_distances is not None:
            return dm.DistanceMatrix(len(self), self.raw_distances, self.ids, "explicit")
        if self.distance_type == "explicit":
            raise ValueError("cannot build distance matrix from coordinates for EXPLICIT type")
        return dm.build(self.ids, self.xy, self.distance_type, ctx)

    def problem(self, ctx=None, build_matrix=False):
        from . import TspProblem
        need = self.raw_distances is not None or self.distance_type != "euc2d" or build_matrix
        return TspProblem(self.ids, self.xy, self.distance_matrix(ctx) if need else None)


def _is_f32(tok):
    try:
        np.float32(tok)
        return True
    except ValueError:
        return False


def _starts_with_number(line):
    t = line.split()
    return bool(t) and _is_f32(t[0])


def read_from_str(text):
    meta, ids, xy, weights = {}, [], [], []
    state, section = "start", None
    for line_no, raw in enumerate(text.splitlines(), start=2):
        line = raw.strip().upper()
        if state == "end":
            break
        if _SECTION.match(line):  # is_state_marker
            if line == "EOF":
                state = "end"
            else:
                state, section = "in", line
            continue
        if state == "start":
            m = _KV.match(line)
            if not m:
                raise ValueError(f"Failed to extract meta data on line.{line_no}")
            meta[m["key"]] = m["val"]
        elif state == "in" and section in ("NODE_COORD_SECTION", "DISPLAY_DATA_SECTION"):
            if not _starts_with_number(line):
                raise ValueError(f"Failed to extract coordinates on line.{line_no}")
            tok = line.split()
            try:
                cid = int(tok[0])
                coords = [np.float32(t) for t in tok[1:]]
            except ValueError:
                raise ValueError(f"Error on line.{line_no} - invalid number") from None
            if len(coords) < 2:
                raise ValueError(f"KDPoint requires at least 2 coordinates, got {len(coords)}")
            ids.append(cid)
            xy.append(coords[:2])
        elif state == "in" and section == "EDGE_WEIGHT_SECTION":
            weights.extend(np.float32(t) for t in line.split() if _is_f32(t))
    if meta.get("TYPE", "").strip() == "ATSP":
        raise ValueError("ATSP (asymmetric TSP) is not supported")
    ewt = meta.get("EDGE_WEIGHT_TYPE", "").strip()
    distance_type = {"EUC_2D": "euc2d", "GEO": "geo", "EXPLICIT": "explicit"}.get(ewt, "euc2d")
    try:
        dimension = int(meta.get("DIMENSION", "0").strip())
    except ValueError:
        dimension = 0
    raw = None
    if weights:
        n = dimension
        w = np.asarray(weights, dtype=np.float32)
        fmt = meta.get("EDGE_WEIGHT_FORMAT", "").strip()
        if fmt == "FULL_MATRIX":
            if len(w) != n * n:
                raise ValueError(f"FULL_MATRIX: expected {n * n} tokens, got {len(w)}")
            full = w.reshape(n, n)
        elif fmt == "UPPER_ROW":
            if len(w) != n * (n - 1) // 2:
                raise ValueError(f"UPPER_ROW: expected {n * (n - 1) // 2} tokens, got {len(w)}")
            full = np.zeros((n, n), dtype=np.float32)
            full[np.triu_indices(n, 1)] = w
            full = full + full.T
        elif fmt == "LOWER_DIAG_ROW":
            if len(w) != n * (n + 1) // 2:
                raise ValueError(f"LOWER_DIAG_ROW: expected {n * (n + 1) // 2} tokens, got {len(w)}")
            full = np.zeros((n, n), dtype=np.float32)
            full[np.tril_indices(n, 0)] = w
        else:
            raise ValueError(f"Unsupported EDGE_WEIGHT_FORMAT: {fmt}")
        raw = np.concatenate([full[i, :i] for i in range(1, n)] or [np.zeros(0, np.float32)]).astype(np.float32)
    if not xy and raw is not None:
        cols = int(np.ceil(np.sqrt(dimension)))
        ids = list(range(1, dimension + 1))
        xy = [[np.float32(i % cols), np.float32(i // cols)] for i in range(dimension)]
    if not xy and raw is None:
        raise ValueError("Found no valid city coordinates")
    return TspLibData(meta.get("NAME", "unspecified").lower(), meta.get("COMMENT", "unspecified").lower(), ids, xy,
                      dimension, raw, distance_type)


def read_from_file(path):
    try:
        with open(path) as fh:
            text = fh.read()
    except OSError:
        raise ValueError("tsplib: failed to read file") from None
    return read_from_str(text)
