"""Minimal TSPLIB reader for the TEST SUITE (test infrastructure, not product code).

Follows the observable behaviour of the reference parser (src/tsp/tsplib.rs:142-255): lines are
trimmed and upper-cased, coordinates are parsed as f32 (`f32::from_str`), EXPLICIT weights are
repacked to the strict lower triangle (tsplib.rs:262-320).  Unsupported EDGE_WEIGHT_TYPEs (ATT)
silently fall back to EUC_2D like the reference (tsplib.rs:199-202).
"""
import numpy as np


def parse_tsplib(path):
    meta, ids, xy, weights = {}, [], [], []
    section = None
    with open(path) as fh:
        for raw in fh:
            line = raw.strip().upper()
            if not line:
                continue
            if line == "EOF":
                break
            if line.replace("_", "").isalpha() and ":" not in line:
                section = line
                continue
            if section is None:
                if ":" in line:
                    k, v = line.split(":", 1)
                    meta[k.strip()] = v.strip()
                continue
            tok = line.split()
            if section in ("NODE_COORD_SECTION", "DISPLAY_DATA_SECTION"):
                ids.append(int(tok[0]))
                xy.append([np.float32(tok[1]), np.float32(tok[2])])
            elif section == "EDGE_WEIGHT_SECTION":
                weights.extend(np.float32(t) for t in tok)
    n = int(meta.get("DIMENSION", len(xy)))
    packed = None
    if weights:
        w = np.asarray(weights, dtype=np.float32)
        fmt = meta.get("EDGE_WEIGHT_FORMAT", "")
        full = np.zeros((n, n), dtype=np.float32)
        if fmt == "FULL_MATRIX":
            full = w.reshape(n, n)
        elif fmt == "UPPER_ROW":
            full[np.triu_indices(n, 1)] = w
            full = full + full.T
        elif fmt == "LOWER_DIAG_ROW":
            full[np.tril_indices(n, 0)] = w
        else:
            raise ValueError(f"Unsupported EDGE_WEIGHT_FORMAT: {fmt}")
        packed = np.concatenate([full[i, :i] for i in range(1, n)]).astype(np.float32)
    if not xy and packed is not None:  # tsplib.rs:251-258 grid placeholder coords
        cols = int(np.ceil(np.sqrt(n)))
        ids = list(range(1, n + 1))
        xy = [[np.float32(i % cols), np.float32(i // cols)] for i in range(n)]
    ewt = meta.get("EDGE_WEIGHT_TYPE", "EUC_2D")
    return {
        "name": meta.get("NAME", "").lower(),
        "n": len(xy),
        "ids": np.asarray(ids, dtype=np.int64),
        "xy": np.ascontiguousarray(np.asarray(xy, dtype=np.float32)),
        "packed": packed,
        "edge_weight_type": ewt,
    }


def parse_opt_tour(path):
    ids, on = [], False
    with open(path) as fh:
        for raw in fh:
            line = raw.strip().upper()
            if line == "TOUR_SECTION":
                on = True
                continue
            if not on:
                continue
            if line in ("-1", "EOF", ""):
                if line == "-1":
                    break
                continue
            ids.extend(int(t) for t in line.split())
    return ids
