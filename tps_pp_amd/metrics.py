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
