"""Text-recognition metrics of the evaluation harness (host side, pure Python).

Mirror of `mmocr/core/evaluation/ocr_metric.py:8-134` (`cal_true_positive_char`, `count_matches`,
`eval_ocr_metric`): word accuracy (exact / ignore case / ignore case and symbols), character
recall / precision over `difflib` matching blocks, 1 - normalised edit distance.  The reference takes
the Levenshtein distance from rapidfuzz (`string_metric.levenshtein`, unit costs); it is restated
here as the textbook two-row dynamic programme so the harness has no third-party dependency.
"""
import re
from difflib import SequenceMatcher

_SYMBOLS = re.compile("[^A-Z^a-z^0-9^一-龥]")


def levenshtein(a: str, b: str) -> int:
    """Edit distance with unit insert / delete / substitute costs."""
    if a == b:
        return 0
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def cal_true_positive_char(pred: str, gt: str) -> int:
    return sum(e2 - s2 for op, _, _, s2, e2 in SequenceMatcher(None, pred, gt).get_opcodes() if op == "equal")


def count_matches(pred_texts, gt_texts):
    res = dict(gt_char_num=0, pred_char_num=0, true_positive_char_num=0, gt_word_num=0, match_word_num=0,
               match_word_ignore_case=0, match_word_ignore_case_symbol=0)
    ned_sum = 0.0
    for pred, gt in zip(pred_texts, gt_texts):
        res["match_word_num"] += int(gt == pred)
        gl, pl = gt.lower(), pred.lower()
        res["match_word_ignore_case"] += int(gl == pl)
        gi, pi = _SYMBOLS.sub("", gl), _SYMBOLS.sub("", pl)
        res["match_word_ignore_case_symbol"] += int(gi == pi)
        res["gt_word_num"] += 1
        ned_sum += float(levenshtein(pi, gi)) / max(1, len(gi), len(pi))
        res["gt_char_num"] += len(gi)
        res["pred_char_num"] += len(pi)
        res["true_positive_char_num"] += cal_true_positive_char(pi, gi)
    res["ned"] = ned_sum / max(1, len(gt_texts))
    return res


def eval_ocr_metric(pred_texts, gt_texts, all_metrics=False):
    """`eval_ocr_metric`: the reference reports only `word_acc_ignore_case_symbol` (its other entries are
    commented out, ocr_metric.py:124-129); `all_metrics=True` returns the full upstream set."""
    assert isinstance(pred_texts, list) and isinstance(gt_texts, list) and len(pred_texts) == len(gt_texts)
    m = count_matches(pred_texts, gt_texts)
    eps = 1e-8
    out = {"word_acc_ignore_case_symbol": m["match_word_ignore_case_symbol"] / (eps + m["gt_word_num"])}
    if all_metrics:
        out.update({"word_acc": m["match_word_num"] / (eps + m["gt_word_num"]),
                    "word_acc_ignore_case": m["match_word_ignore_case"] / (eps + m["gt_word_num"]),
                    "char_recall": m["true_positive_char_num"] / (eps + m["gt_char_num"]),
                    "char_precision": m["true_positive_char_num"] / (eps + m["pred_char_num"]),
                    "1-N.E.D": 1.0 - m["ned"]})
    return {k: float("{:.4f}".format(v)) for k, v in out.items()}


def precision_agreement(model, img, img_metas, mode):
    """How far a reduced-precision configuration of a recogniser (`mode`: torch.bfloat16 or "bf16x3", see
    `EncodeDecodeRecognizer.set_compute_dtype`, or a dict with one such mode per stage: backbone / encoder / decoder) moves its decisions away from the exact-fp32 kernels of the SAME model on
    the same images (BASELINE.json configs[4], "word-accuracy parity check"):
      * teacher forcing: the reduced-precision model is fed the fp32 run's greedy tokens; fraction of positions (up to and
        including the fp32 run's <EOS>) whose arg-max is the fp32 one -- every position is an independent decision;
      * greedy: word agreement (identical strings) and character agreement (difflib matching blocks over the fp32
        strings' characters) -- on random-init weights one flipped near-tie rewrites the rest of a string;
      * the same two restricted to decisions the fp32 run is sure of (top-1 minus top-2 score >= 0.05): positions for the
        teacher-forced figure; for the greedy one, every string compared up to the first position the fp32 run is not sure of.
    Returns a dict of floats; `model` is left in the exact-fp32 configuration."""
    import torch
    conv, dec = model.label_convertor, model.decoder
    metas = [dict(m) for m in img_metas]
    for m in metas:
        m.setdefault("valid_ratio", 1.0 * m["resize_shape"][1] / img.size(-1))

    def run(forced=None):
        feat = model.extract_feat(img, test=True)
        feat = feat["output"] if isinstance(feat, dict) else feat
        out_enc = model.encoder(feat, metas)
        if forced is None:
            return dec(feat, out_enc, None, metas, train_mode=False)
        return dec(feat, out_enc, dict(padded_targets=forced), metas, train_mode=True)

    def apply(md):
        """`md`: one mode for every stage, or a dict per stage (keys backbone / encoder / decoder; missing = fp32) --
        the per-stage form is how bench.py locates which stage's bf16 arithmetic flips decisions."""
        if not isinstance(md, dict):
            model.set_compute_dtype(md)
            return
        model.set_compute_dtype(None)
        bb = md.get("backbone")
        model.backbone.compute_dtype = bb
        if model.tpsnet is not None:
            model.tpsnet.compute_dtype = bb if bb == "bf16x3" else None
        if model.encoder is not None and hasattr(model.encoder, "compute_dtype"):
            model.encoder.compute_dtype = md.get("encoder")
        if model.decoder is not None and hasattr(model.decoder, "compute_dtype"):
            model.decoder.compute_dtype = md.get("decoder")

    with torch.no_grad():
        model.set_compute_dtype(None)
        ref = run()                                                   # (N, L, num_classes - 1) softmax scores
        ref_tok = ref.argmax(-1)                                      # (N, L)
        ref_txt = conv.idx2str(conv.tensor2idx(ref)[0])
        n, L = ref_tok.shape
        start = torch.full((n, 1), dec.start_idx, dtype=ref_tok.dtype, device=ref_tok.device)
        forced = torch.cat([start, ref_tok[:, :L - 1]], dim=1)        # position t sees <BOS>, tok_0 .. tok_{t-1}
        self_tf = run(forced).argmax(-1)                              # sanity: the fp32 kernels reproduce themselves
        apply(mode)
        low_tf = run(forced).argmax(-1)
        low = run()
        low_txt = conv.idx2str(conv.tensor2idx(low)[0])
        model.set_compute_dtype(None)
    is_end = ref_tok == conv.end_idx
    first_end = torch.where(is_end.any(1), is_end.float().argmax(1), torch.full((n,), L - 1, device=ref_tok.device))
    valid = torch.arange(L, device=ref_tok.device)[None, :] <= first_end[:, None]
    m = count_matches(low_txt, ref_txt)
    # decisions the fp32 run itself is sure of: top-1 minus top-2 softmax score >= 0.05.  Random-init weights put many
    # positions on near-ties, where ANY rounding flips the arg-max; restricted to confident positions / words the figure
    # says something about the arithmetic without a trained checkpoint
    top2 = ref.topk(2, dim=-1).values
    confident = ((top2[..., 0] - top2[..., 1]) >= 0.05) & valid
    # greedy decoding up to the first position the fp32 run is NOT sure of: behind a flipped near-tie the two runs decode
    # different prefixes and nothing can be compared any more (whole 40-position random-init words are practically never
    # confident everywhere, so a per-word restriction would select nothing)
    low_tok = low.argmax(-1)
    unsure = valid & ~confident
    first_unsure = torch.where(unsure.any(1), unsure.float().argmax(1), first_end + 1)
    prefix = torch.arange(L, device=ref_tok.device)[None, :] < first_unsure[:, None]
    prefix_ok = ((low_tok == ref_tok) | ~prefix).all(1)
    return {"images": n, "positions": int(valid.sum()),
            "positions_margin_ge_0.05": int(confident.sum()),
            "teacher_forced_agreement_margin_ge_0.05":
                float((low_tf == ref_tok)[confident].float().mean()) if bool(confident.any()) else float("nan"),
            "greedy_agreement_up_to_first_margin_lt_0.05": float(prefix_ok.float().mean()),
            "mean_confident_prefix_length": float(first_unsure.float().mean()),
            "teacher_forced_argmax_agreement": float((low_tf == ref_tok)[valid].float().mean()),
            "teacher_forced_self_check_fp32": float((self_tf == ref_tok)[valid].float().mean()),
            "greedy_word_agreement": sum(a == b for a, b in zip(low_txt, ref_txt)) / max(1, n),
            "greedy_char_agreement": m["true_positive_char_num"] / max(1, m["gt_char_num"]),
            "mean_fp32_string_length": sum(len(t) for t in ref_txt) / max(1, n)}
