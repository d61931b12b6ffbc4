"""Nested python records (dicts / lists of basic types and numpy arrays, e.g. PlanEnv.serialize()) <-> an .npz fixture:
arrays are stored as npz members, everything else in a JSON skeleton.  TEST INFRASTRUCTURE (fixtures only)."""
import json

import numpy as np


def pack(obj):
    """-> (skeleton_json, {member_name: array})"""
    arrays = {}

    def walk(o):
        if isinstance(o, np.ndarray):
            key = "arr%04d" % len(arrays)
            arrays[key] = o
            return {"__array__": key}
        if isinstance(o, dict):
            assert all(isinstance(k, str) for k in o), "record keys must be strings"
            return {"__dict__": [[k, walk(v)] for k, v in o.items()]}
        if isinstance(o, (list, tuple)):
            return {"__list__": [walk(v) for v in o], "tuple": isinstance(o, tuple)}
        if isinstance(o, (np.floating,)):
            return {"__f64__": float(o)}
        if isinstance(o, (np.integer,)):
            return {"__int__": int(o)}
        if isinstance(o, (np.bool_,)):
            return {"__bool__": bool(o)}
        if o is None or isinstance(o, (bool, int, float, str)):
            return o
        raise TypeError("cannot pack %r" % type(o))

    return json.dumps(walk(obj)), arrays


def unpack(skeleton_json, arrays):
    def walk(o):
        if isinstance(o, dict):
            if "__array__" in o:
                return np.array(arrays[o["__array__"]])
            if "__dict__" in o:
                return dict((k, walk(v)) for k, v in o["__dict__"])
            if "__list__" in o:
                items = [walk(v) for v in o["__list__"]]
                return tuple(items) if o.get("tuple") else items
            if "__f64__" in o:
                return np.float64(o["__f64__"])
            if "__int__" in o:
                return o["__int__"]
            if "__bool__" in o:
                return o["__bool__"]
        return o

    return walk(json.loads(str(skeleton_json)))


def same(a, b, path="record"):
    """Key-for-key, value-for-value comparison of two records; returns a list of differences (empty = equal)."""
    diffs = []
    if isinstance(a, dict) or isinstance(b, dict):
        if not (isinstance(a, dict) and isinstance(b, dict)):
            return ["%s: %s vs %s" % (path, type(a).__name__, type(b).__name__)]
        if list(sorted(a)) != list(sorted(b)):
            diffs.append("%s: keys %s vs %s" % (path, sorted(set(a) - set(b)), sorted(set(b) - set(a))))
        for k in a:
            if k in b:
                diffs += same(a[k], b[k], path + "/" + k)
        return diffs
    if isinstance(a, (list, tuple)) or isinstance(b, (list, tuple)):
        if not (isinstance(a, (list, tuple)) and isinstance(b, (list, tuple))) or len(a) != len(b):
            return ["%s: sequence %r vs %r" % (path, a, b)]
        for k, (x, y) in enumerate(zip(a, b)):
            diffs += same(x, y, "%s[%d]" % (path, k))
        return diffs
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        a, b = np.asarray(a), np.asarray(b)
        if a.shape != b.shape or a.dtype.kind != b.dtype.kind or not np.array_equal(a, b):
            return ["%s: arrays differ (%s %s vs %s %s)" % (path, a.dtype, a.shape, b.dtype, b.shape)]
        return []
    if isinstance(a, bool) != isinstance(b, bool) and not (isinstance(a, (bool, np.bool_)) and isinstance(b, (bool, np.bool_))):
        return ["%s: %r vs %r" % (path, a, b)]
    if a != b:
        return ["%s: %r vs %r" % (path, a, b)]
    return []
