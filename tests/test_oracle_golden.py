"""CPU oracle vs the reference's own recorded outputs (tests/golden/ref_quantize.json) and the
known answers probed on the reference's quantizer (js/renderer.js:59-72)."""
import base64
import json
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def ref(golden_dir):
    with open(os.path.join(golden_dir, "ref_quantize.json")) as fh:
        return json.load(fh)


def _case_arrays(case):
    rgba = np.frombuffer(base64.b64decode(case["rgba"]), dtype=np.uint8).copy()
    chars = np.frombuffer(base64.b64decode(case["chars_u16"]), dtype=np.uint16)
    return rgba, chars


def test_quantize_matches_compositeAndQuantize(oracle, ref):
    assert len(ref["cases"]) >= 18
    for case in ref["cases"]:
        rgba, want = _case_arrays(case)
        cols, rows = case["cols"], case["rows"]
        if "ui" in case:
            # the reference's UI layer wins and is mirrored into alpha (js/renderer.js:40-47);
            # express the same thing through the alpha-override channel the ASCII stage reads
            ui = np.asarray(case["ui"], dtype=np.uint8)
            a = rgba.reshape(-1, 4)
            a[:, 3] = np.where(ui > 0, ui, 1)
        else:
            rgba.reshape(-1, 4)[:, 3] = 255
        got = oracle.quantize(rgba, cols, rows, case["ramp"], mode_on=0)
        assert np.array_equal(got.reshape(-1).astype(np.uint16), want), (case["name"], case["ramp"])


def _with_ui_alpha(case, rgba):
    if "ui" in case:
        ui = np.asarray(case["ui"], dtype=np.uint8)
        rgba.reshape(-1, 4)[:, 3] = np.where(ui > 0, ui, 1)
    else:
        rgba.reshape(-1, 4)[:, 3] = 255
    return rgba


def test_text_colours_match_compositeAndQuantize(oracle, ref):
    """dataBuffer_colors of the reference (js/renderer.js:74-84) for every recorded grid and ramp; the
    number of levels per channel follows the ramp length as in renderer.js:13-19."""
    seen = set()
    for case in ref["cases"]:
        rgba, _ = _case_arrays(case)
        want = np.frombuffer(base64.b64decode(case["colors_u32"]), dtype=np.uint32)
        got = oracle.quantize_colors(_with_ui_alpha(case, rgba), case["cols"], case["rows"], len(case["ramp"]))
        assert np.array_equal(got.reshape(-1), want), (case["name"], case["ramp"])
        seen.add(max(3, len(case["ramp"])))
        assert not oracle.quantize_colors(rgba, case["cols"], case["rows"], len(case["ramp"]), grayscale=True).any()
    assert len(seen) >= 4          # several step counts are covered (3, 10, 12, 70)


def test_alpha_written_by_reference_is_the_override_convention(ref):
    for case in ref["cases"]:
        if "ui" not in case:
            continue
        ui = np.asarray(case["ui"], dtype=np.uint8)
        after = np.frombuffer(base64.b64decode(case["alpha_after_u8"]), dtype=np.uint8)
        assert np.array_equal(after, np.where(ui > 0, ui, 1))


def test_integer_identity_for_every_sum_and_ramp_length(oracle):
    """idx = floor((2*s*(L-1) + 765) / 1530) equals the reference's rounded float formula."""
    s = np.arange(766)
    rgba = np.zeros((766, 4), dtype=np.uint8)
    rgba[:, 0] = np.minimum(255, s)
    rgba[:, 1] = np.minimum(255, s - rgba[:, 0])
    rgba[:, 2] = s - rgba[:, 0].astype(int) - rgba[:, 1]
    rgba[:, 3] = 255
    for L in list(range(1, 72)) + [128, 255]:
        ramp = bytes((33 + k) % 256 for k in range(L))
        got = oracle.quantize(rgba, 766, 1, ramp, mode_on=0).reshape(-1)
        idx = (2 * s * (L - 1) + 765) // 1530
        assert np.array_equal(got, np.frombuffer(ramp, dtype=np.uint8)[idx]), L


@pytest.mark.parametrize("rgb,code", [((0, 0, 0), 64), ((255, 255, 255), 32), ((128, 128, 128), 61), ((42, 43, 43), 35)])
def test_known_answers(oracle, rgb, code):
    rgba = np.array([[*rgb, 255]], dtype=np.uint8)
    assert oracle.quantize(rgba, 1, 1, "@%#*+=-:. ")[0, 0] == code


def test_override_window(oracle):
    for a, is_override in [(0, False), (1, False), (2, True), (126, True), (254, True), (255, False)]:
        rgba = np.array([[255, 255, 255, a]], dtype=np.uint8)
        got = oracle.quantize(rgba, 1, 1, "@%#*+=-:. ")[0, 0]
        assert got == (a if is_override else 32)


def _modal_reference(rgba, cols, rows, ramp, radius, thresh):
    """Independent pure-Python restatement of ascii_pass_shader.js:77-138,169-185 (small grids)."""
    L = len(ramp)
    px = rgba.reshape(rows, cols, 4).astype(int)
    base = (2 * px[:, :, :3].sum(axis=2) * (L - 1) + 765) // 1530
    out = np.zeros((rows, cols), dtype=np.uint8)
    radius = max(1, min(3, radius))
    for y in range(rows):
        for x in range(cols):
            a = px[y, x, 3]
            if 2 <= a <= 254:
                out[y, x] = a
                continue
            neigh = []
            for dy in range(-radius, radius + 1):
                for dx in range(-radius, radius + 1):
                    if dx == 0 and dy == 0:
                        continue
                    nx, ny = min(max(x + dx, 0), cols - 1), min(max(y + dy, 0), rows - 1)
                    if 2 <= px[ny, nx, 3] <= 254:
                        continue
                    neigh.append(base[ny, nx])
            cand, cnt = -1, 0
            for n in neigh:
                if cnt == 0:
                    cand, cnt = n, 1
                elif n == cand:
                    cnt += 1
                else:
                    cnt -= 1
            votes = sum(1 for n in neigh if n == cand) if cand >= 0 else 0
            fin = cand if (cand >= 0 and votes >= thresh and cand != base[y, x]) else base[y, x]
            out[y, x] = ord(ramp[fin])
    return out


@pytest.mark.parametrize("cols,rows,radius,thresh", [(17, 9, 2, 12), (5, 4, 1, 5), (9, 9, 3, 20), (1, 1, 2, 12), (3, 1, 2, 1)])
def test_modal_filter_against_python_restatement(oracle, cols, rows, radius, thresh):
    rng = np.random.default_rng(cols * 100 + rows)
    # few distinct grey levels so majorities actually form
    lv = rng.choice([0, 60, 120, 250], size=(rows * cols, 1), p=[0.55, 0.2, 0.15, 0.1]).astype(np.uint8)
    rgba = np.concatenate([lv, lv, lv, np.full_like(lv, 255)], axis=1)
    ov = rng.random(rows * cols) < 0.1
    rgba[ov, 3] = rng.integers(2, 255, size=ov.sum())
    ramp = "@%#*+=-:. "
    got = oracle.quantize(rgba, cols, rows, ramp, mode_on=1, mode_radius=radius, mode_thresh=thresh)
    want = _modal_reference(rgba, cols, rows, ramp, radius, thresh)
    assert np.array_equal(got, want)
    if cols * rows > 50:
        assert not np.array_equal(got, oracle.quantize(rgba, cols, rows, ramp, mode_on=0))
