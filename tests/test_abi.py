"""The C-ABI library loads on a machine without a GPU and exports every symbol include/stv.h declares."""
from __future__ import annotations

import ctypes
import os
import re

from style_transfer_visualizer_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols() -> set[str]:
    text = open(os.path.join(ROOT, "include", "stv.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(stv_[a-z0-9_]+)\s*\(", text))


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/stv.h but not exported"
    assert declared == set(_lib.SIGNATURES), "ctypes SIGNATURES out of sync with include/stv.h"


def test_version_and_sizing_helpers_run_without_gpu():
    lib = _lib.load()
    assert lib.stv_version() >= 100
    # pure host arithmetic, no device work
    assert lib.stv_gram_ksplit(1 << 20, 64) >= 1
    assert lib.stv_gram_partials_bytes(4096, 512) == lib.stv_gram_ksplit(4096, 512) * 512 * 512 * 4
    assert lib.stv_lbfgs_workspace_bytes(3 * 64 * 64, 100) > 2 * 101 * 3 * 64 * 64 * 4
    assert lib.stv_lbfgsc_workspace_bytes(3 * 64 * 64, 100) > 2 * 101 * 3 * 64 * 64 * 4
    assert lib.stv_gram_loss_parts(64) * 128 >= 64 * 64   # one loss partial per 128 Gram elements


def test_op_struct_layout_matches_header():
    # stv_op_t: 8 int32, 1 int64, 4 float, 8 pointers
    assert ctypes.sizeof(_lib.StvOp) == 8 * 4 + 8 + 4 * 4 + 8 * 8


def test_bad_arguments_are_rejected_not_executed():
    lib = _lib.load()
    # null pointers -> STV_ERR_ARG before any HIP call
    assert lib.stv_conv_igemm(None, None, None, None, None, 8, 8, 16, 16, 9, 0, 0, None) == 1
    assert lib.stv_gram_partial(None, None, 10, 64, 0, None) == 1
    assert lib.stv_lbfgs_step(None, None, None, None, 10, 100, 0, 1.0, 1e-7, 1e-9, None) == 1


def test_tile_table_round_trip_and_modes(monkeypatch):
    """The persisted tile choices reach the library at load time (host-only calls, no GPU): export returns what
    conv_tiles_gfx950.json holds; a shape's tile is the table's unless STV_CONV_TUNE=0 asks for the analytic one;
    malformed entries are rejected, an empty import clears the table."""
    import json

    lib = _lib.load()
    doc = json.load(open(_lib.TILE_TABLE_PATH))
    entries = _lib.export_tile_table()
    want = {(e["H"], e["W"], e["cin"], e["cout"], e["taps"], e["elem_bytes"]): e["cfg"] for e in doc["entries"]}
    got = {(e["H"], e["W"], e["cin"], e["cout"], e["taps"], e["elem_bytes"]): e["cfg"] for e in entries}
    assert want and all(got.get(k) == v for k, v in want.items())
    differing = [e for e in doc["entries"] if e["cfg"] != e["analytic"] and e["taps"] == 9 and e["elem_bytes"] == 2]
    assert differing, "the measured table should differ from the analytic choice somewhere"
    e = differing[0]
    monkeypatch.delenv("STV_CONV_TUNE", raising=False)
    assert lib.stv_conv_config(e["H"], e["W"], e["cin"], e["cout"], 9, _lib.STV_BF16) == e["cfg"]
    monkeypatch.setenv("STV_CONV_TUNE", "0")
    assert lib.stv_conv_config(e["H"], e["W"], e["cin"], e["cout"], 9, _lib.STV_BF16) == e["analytic"]
    bad = (ctypes.c_int * 7)(64, 64, 512, 512, 9, 3, 4)          # element size 3
    assert lib.stv_conv_tune_import(bad, 1) == 1
    n_before = lib.stv_conv_tune_export(None, 0)
    assert lib.stv_conv_tune_import(None, 0) == 0 and lib.stv_conv_tune_export(None, 0) == 0
    _lib._import_tile_table(lib)                                  # restore for the tests that follow
    assert lib.stv_conv_tune_export(None, 0) == n_before


def test_a_table_the_library_rejects_is_ignored_not_fatal(monkeypatch, tmp_path):
    """A tile table with an entry the library cannot take (a tile index that no longer exists), a malformed file or a
    table measured on another architecture must not make the package unusable: a warning, an empty table, the
    analytic choices (load() used to raise)."""
    import json

    import pytest

    lib = _lib.load()
    good = json.load(open(_lib.TILE_TABLE_PATH))
    n_before = lib.stv_conv_tune_export(None, 0)
    e = dict(good["entries"][0])
    for name, doc in (("bad_cfg", {"entries": [dict(e, cfg=999)]}), ("bad_key", {"entries": [{"H": 1}]}), ("not_json", None)):
        path = tmp_path / f"{name}.json"
        path.write_text("{" if doc is None else json.dumps(doc))
        monkeypatch.setattr(_lib, "TILE_TABLE_PATH", str(path))
        with pytest.warns(RuntimeWarning, match="ignored"):
            _lib._import_tile_table(lib)
        assert lib.stv_conv_tune_export(None, 0) == 0 and _lib.tile_table_info["entries"] == 0
        assert lib.stv_conv_config(e["H"], e["W"], e["cin"], e["cout"], e["taps"], _lib.STV_BF16) == e["analytic"]
    other = tmp_path / "other_arch.json"
    other.write_text(json.dumps(dict(good, arch="gfx942")))
    monkeypatch.setattr(_lib, "TILE_TABLE_PATH", str(other))
    monkeypatch.setattr(_lib, "_device_arch", lambda: "gfx950")
    with pytest.warns(RuntimeWarning, match="gfx942"):
        _lib._import_tile_table(lib)
    assert lib.stv_conv_tune_export(None, 0) == 0
    monkeypatch.undo()
    _lib._import_tile_table(lib)                                  # restore for the tests that follow
    assert lib.stv_conv_tune_export(None, 0) == n_before and _lib.tile_table_info["entries"] == n_before


def test_weight_stationary_kernel_selection_policy(monkeypatch):
    """Which 3x3 launches take the weight-stationary kernel (host-side decision, no GPU needed: stv_conv_uses_ws): the
    64 -> 64 layers always; 128 -> 128 only in the backward form (mask / Gram term) from 8 tiles of 2 x 32 pixels per
    workgroup up - where it measured faster; STV_CONV_WS128 = 0 / 2 override; fp32 never."""
    lib = _lib.load()
    for var in ("STV_CONV_WS", "STV_CONV_WS128", "STV_CONV_CFG"):
        monkeypatch.delenv(var, raising=False)
    BF16, F32, MASK, RELU = _lib.STV_BF16, _lib.STV_F32, _lib.MASK, _lib.RELU_IN | _lib.RELU_OUT

    def uses(H, cin, cout, dtype=BF16, flags=0, has_ref=0, has_pool=0):
        return bool(lib.stv_conv_uses_ws(H, H, cin, cout, 9, dtype, flags | _lib.W_BLOCKED, has_ref, has_pool))
    assert uses(1024, 64, 64, flags=RELU, has_pool=1) and uses(512, 64, 64, flags=MASK, has_ref=1)
    assert not uses(512, 64, 128) and not uses(512, 64, 64, dtype=F32)
    assert uses(512, 128, 128, flags=MASK, has_ref=1)              # conv2_2's backward at 1024^2: 4,096 tiles on 256 CUs
    assert not uses(256, 128, 128, flags=MASK, has_ref=1)          # ... at 512^2: 4 tiles per workgroup
    assert not uses(512, 128, 128, flags=RELU, has_pool=1)         # the forward form is no faster than the general kernel
    monkeypatch.setenv("STV_CONV_WS128", "2")
    assert uses(256, 128, 128, flags=MASK, has_ref=1) and uses(512, 128, 128, flags=RELU, has_pool=1)
    assert not uses(512, 128, 256)
    monkeypatch.setenv("STV_CONV_WS128", "0")
    assert not uses(512, 128, 128, flags=MASK, has_ref=1)
    monkeypatch.delenv("STV_CONV_WS128")
    monkeypatch.setenv("STV_CONV_WS", "0")
    assert not uses(1024, 64, 64, flags=RELU, has_pool=1)


def test_bench_names_every_tile_configuration():
    """bench.py maps a tile index to the kernel instantiation rocprofv3 reports: its tables must cover every tile the
    library can choose (a new tile once crashed the bench with a KeyError)."""
    import bench
    n = _lib.load().stv_conv_num_configs()
    assert sorted(bench._CFG_NAMES) == list(range(n)) and sorted(bench._CFG_KS) == list(range(n))
