"""Integer geometry of the four processors (oracle side; test infrastructure only).

Every function cites the reference lines it follows.  All results are plain Python
ints so they can be compared bit-exactly with `advx_plan_describe`.
"""
import math

import numpy as np


# ----------------------------------------------------------------------------- mllama
def mllama_supported_arrangements(max_image_tiles):
    """All (a, b) with a*b <= max tiles, a-major order.

    Restates transformers 5.15 `image_processing_pil_mllama.get_all_supported_aspect_ratios`
    (the helper `llama32processor.py:262` reaches through `get_optimal_tiled_canvas`).
    """
    out = []
    for a in range(1, max_image_tiles + 1):
        for b in range(1, max_image_tiles + 1):
            if a * b <= max_image_tiles:
                out.append((a, b))
    return out


def mllama_optimal_canvas(image_height, image_width, max_image_tiles, tile_size):
    """Canvas (height, width) in pixels. Follows `get_optimal_tiled_canvas` (transformers
    5.15 `image_processing_pil_mllama.py:299`), called at llama32processor.py:262-267.

    Float64 scale per candidate = min(canvas_h / H, canvas_w / W); prefer the smallest
    scale >= 1, else the largest scale < 1; ties -> smallest canvas area (first wins).
    """
    arr = mllama_supported_arrangements(max_image_tiles)
    cands = [(a * tile_size, b * tile_size) for (a, b) in arr]
    scales = []
    for (ch, cw) in cands:
        sh = ch / image_height
        sw = cw / image_width
        scales.append(sh if sw > sh else sw)
    ups = [s for s in scales if s >= 1]
    if ups:
        sel = min(ups)
    else:
        sel = max(s for s in scales if s < 1)
    chosen = [c for c, s in zip(cands, scales) if s == sel]
    best = chosen[0]
    for c in chosen[1:]:
        if c[0] * c[1] < best[0] * best[1]:
            best = c
    return int(best[0]), int(best[1])


def mllama_fit_to_canvas(image_height, image_width, canvas_height, canvas_width, tile_size):
    """Resized (height, width). Follows `get_image_size_fit_to_canvas`
    (`image_processing_pil_mllama.py:246`), called at llama32processor.py:271-277."""
    target_w = min(max(image_width, tile_size), canvas_width)
    target_h = min(max(image_height, tile_size), canvas_height)
    scale_h = target_h / image_height
    scale_w = target_w / image_width
    if scale_w < scale_h:
        new_w = target_w
        new_h = min(math.floor(image_height * scale_w) or 1, target_h)
    else:
        new_h = target_h
        new_w = min(math.floor(image_width * scale_h) or 1, target_w)
    return int(new_h), int(new_w)


def mllama_geometry(image_height, image_width, max_image_tiles=4, tile_size=560):
    """llama32processor.py:255-279 (`_optimal_size`): (new_h, new_w, tiles_h, tiles_w)."""
    ch, cw = mllama_optimal_canvas(image_height, image_width, max_image_tiles, tile_size)
    nh, nw = mllama_fit_to_canvas(image_height, image_width, ch, cw, tile_size)
    return nh, nw, ch // tile_size, cw // tile_size


def mllama_aspect_ratio_id(tiles_h, tiles_w, max_image_tiles=4):
    """1-based index into the supported arrangements (`convert_aspect_ratios_to_ids`)."""
    return mllama_supported_arrangements(max_image_tiles).index((tiles_h, tiles_w)) + 1


# ------------------------------------------------------------------------------- qwen
def qwen_smart_resize(height, width, patch_size=14, merge_size=2,
                      min_pixels=56 * 56, max_pixels=28 * 28 * 1280):
    """qwen2VLprocessor.py:176-197 (`_optimal_size`): (h_bar, w_bar).

    Python `round` is round-half-to-even on the float64 quotient, as in the reference.
    """
    factor = patch_size * merge_size
    h_bar = round(height / factor) * factor
    w_bar = round(width / factor) * factor
    if h_bar * w_bar > max_pixels:
        beta = math.sqrt((height * width) / max_pixels)
        h_bar = math.floor(height / beta / factor) * factor
        w_bar = math.floor(width / beta / factor) * factor
    elif h_bar * w_bar < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        h_bar = math.ceil(height * beta / factor) * factor
        w_bar = math.ceil(width * beta / factor) * factor
    return int(h_bar), int(w_bar)


def qwen_patch_index(c, t, y, x, grid_w, patch_size=14, merge_size=2, temporal=2):
    """(row, col) of canvas element (channel c, temporal copy t, y, x) in the
    [grid_h*grid_w, C*temporal*patch*patch] output of qwen2VLprocessor.py:249-267
    (reshape to 9-D, permute(0,3,6,4,7,2,1,5,8), flatten); grid_t == 1."""
    gy, ph = divmod(y, patch_size)
    gx, pw = divmod(x, patch_size)
    by, mh = divmod(gy, merge_size)
    bx, mw = divmod(gx, merge_size)
    row = ((by * (grid_w // merge_size) + bx) * merge_size + mh) * merge_size + mw
    col = ((c * temporal + t) * patch_size + ph) * patch_size + pw
    return row, col


# ------------------------------------------------------------------------------- phi3
def phi3_hd_geometry(height, width, hd_num=6):
    """phi3processor.py:173-216 (`_pad`): geometry of the HD transform.

    Returns dict with: trans (bool), new_h/new_w of the 2-tap bilinear resize in the
    (possibly transposed) frame, pad_top/pad_bottom in that frame, and the final
    (h, w) of the padded image in the ORIGINAL orientation.
    """
    trans = False
    if width < height:
        trans = True
        height, width = width, height
    ratio = width / height
    scale = 1
    while scale * np.ceil(scale / ratio) <= hd_num:
        scale += 1
    scale -= 1
    new_w = int(scale * 336)
    new_h = int(new_w / ratio)
    target_h = int(np.ceil(new_h / 336) * 336)
    pad_top = (target_h - new_h) // 2
    pad_bottom = target_h - new_h - pad_top
    out_h, out_w = (new_w, target_h) if trans else (target_h, new_w)
    return dict(trans=trans, new_h=int(new_h), new_w=int(new_w), pad_top=int(pad_top),
                pad_bottom=int(pad_bottom), out_h=int(out_h), out_w=int(out_w))


def phi3_num_img_tokens(h, w):
    """phi3processor.py:244."""
    return int(((h // 336) * (w // 336) + 1) * 144 + 1 + (h // 336 + 1) * 12)


# ------------------------------------------------------------------------------ masks
def mask_box(mask_type, mask_size, H, W):
    """attack_model.py:66-84 (`create_mask`) as a half-open box (y0, y1, x0, x1) of
    ones; everything else zero.  `random_square` is not covered (Q9: per-step update
    raises in the reference)."""
    if mask_type == "corner":
        n = mask_size
        return 0, min(n, H), 0, min(n, W)
    if mask_type == "bottom_lines":
        k = mask_size
        return max(H - k, 0), H, 0, W
    return 0, H, 0, W
