"""BASELINE.json configs[3]/[4] as a runnable evaluation step: batch-sharded NRTR + TPS++ inference, one
process per GPU, the per-step scores all-gathered over RCCL (the only collective of the path), strings and
word accuracy computed from the gathered tensor.

    python tests/tools/eval_e2e.py --batch 256                       # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        --master-port 29500 tests/tools/eval_e2e.py --batch 2048     # configs[3]: 256 images per GPU

Synthetic IC15-shaped crops (3x32x128, ImageNet normalisation) and seeded random-init weights (there is no
network for datasets or checkpoints; pass --checkpoint for a released .pth).  "Ground truth" for the word
accuracy is the CPU oracle's decoding of the first --check images (the parity check of configs[4]).

Lives under tests/ because it calls the oracle (as the checker); the product never does."""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tps_pp_amd as P  # noqa: E402
from tps_pp_amd import dist as tdist, metrics, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--check", type=int, default=4, help="images decoded by the CPU oracle as well")
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--bf16", action="store_true",
                    help="BASELINE.json configs[4]: backbone / TPS++ convolutions and the head's wide projections on the "
                         "bf16 matrix cores (control points, TPS solve, grid, per-step decoder projections stay fp32)")
    ap.add_argument("--x3", action="store_true",
                    help='"bf16x3": fp32 tensors, three-term bf16 split in the convolutions and wide projections '
                         "(within the 1e-4 bar: strings must equal the CPU oracle's)")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or "RANK" in os.environ:
        dist.init_process_group("nccl", rank=rank, world_size=world)

    torch.manual_seed(11)                                      # same weights on every rank
    model = P.build_detector(dict(
        type="NRTR", backbone=dict(type="ResNetABI_v2_large", arch_settings=[3, 4, 6, 6, 3], strides=[2, 1, 2, 1, 2]),
        tpsnet=dict(type="TPS_PP", variant="ResNet45"), encoder=dict(type="NRTREncoder"),
        decoder=dict(type="NRTRDecoder"), loss=dict(type="TFLoss"),
        label_convertor=dict(type="AttnConvertor", dict_type="DICT90", with_unknown=True), max_seq_len=40)).eval()
    if a.checkpoint:
        sd = torch.load(a.checkpoint, map_location="cpu")
        model.load_state_dict(sd.get("state_dict", sd), strict=False)
    else:
        with torch.no_grad():
            model.decoder.classifier.weight.mul_(8.0)          # random init: keep the arg-max away from ties
    cpu_sds = [{k: v.clone() for k, v in m.state_dict().items()}
               for m in (model.backbone, model.tpsnet, model.encoder, model.decoder)]
    model.to(dev)
    if a.bf16:
        model.backbone.compute_dtype = model.encoder.compute_dtype = model.decoder.compute_dtype = torch.bfloat16
    elif a.x3:
        model.backbone.compute_dtype = model.tpsnet.compute_dtype = "bf16x3"
        model.encoder.compute_dtype = model.decoder.compute_dtype = "bf16x3"

    # IC15-shaped synthetic crops, test_pipeline normalisation (crnn_pp_pipeline.py: mean/std of ImageNet)
    n = a.batch
    raw = (synth.smooth_image((n, 3, 32, 128), "ic15.crops", 15) + 1.0) * 0.5           # [0, 1]
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    img = ((torch.from_numpy(raw) - mean) / std).contiguous()
    widths = [128 if i % 4 else 100 for i in range(n)]

    def decode_local(lo, hi):
        metas = [dict(resize_shape=(32, widths[i], 3)) for i in range(lo, hi)]
        x = img[lo:hi].to(dev)
        with torch.no_grad():
            for m in metas:
                m["valid_ratio"] = 1.0 * m["resize_shape"][1] / x.size(-1)
            feat = model.extract_feat(x, test=True)["output"]
            out_enc = model.encoder(feat, metas)
            return model.decoder(feat, out_enc, None, metas, train_mode=False)

    res = tdist.recognize_sharded(decode_local, n, model.label_convertor)       # warm-up + result
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        res = tdist.recognize_sharded(decode_local, n, model.label_convertor)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    if rank == 0 and a.bf16:
        # random-init weights decode 40 characters of noise autoregressively: one flipped arg-max rewrites the rest of
        # the string.  The per-position figure: feed the fp32 model's own tokens (teacher forcing) to the bf16 model and
        # count the positions whose arg-max is still the fp32 token.
        k2 = min(64, n)
        metas = [dict(resize_shape=(32, widths[i], 3), valid_ratio=widths[i] / 128) for i in range(k2)]
        x = img[:k2].to(dev)
        with torch.no_grad():
            model.backbone.compute_dtype = model.encoder.compute_dtype = model.decoder.compute_dtype = None
            f32 = model.extract_feat(x, test=True)["output"]
            model.decoder(f32, model.encoder(f32, metas), None, metas, train_mode=False)
            tok = model.decoder.last_tokens.clone()                                   # (k2, 41): <start>, 40 predictions
            model.backbone.compute_dtype = model.encoder.compute_dtype = model.decoder.compute_dtype = torch.bfloat16
            f16 = model.extract_feat(x, test=True)["output"]
            logits = model.decoder(f16, model.encoder(f16, metas), dict(padded_targets=tok[:, :40]), metas, train_mode=True)
        agree = (logits.argmax(-1) == tok[:, 1:41].to(logits.device)).float().mean().item()
        print(f"bf16 vs fp32 (HIP), teacher-forced on {k2} images: {100 * agree:.2f} % of the {k2 * 40} positions keep "
              f"their arg-max")
    if rank == 0:
        from oracle import tpspp_oracle as TO                  # checker only
        k = min(a.check, n)
        want = TO.recognizer_simple_test(cpu_sds[0], cpu_sds[1], cpu_sds[2], cpu_sds[3], img[:k].numpy(), widths[:k])["text"]
        got = [r["text"] for r in res[:k]]
        acc = metrics.eval_ocr_metric(got, want, all_metrics=True)
        print(f"ranks {world}, batch {n}, {'bf16' if a.bf16 else 'bf16x3' if a.x3 else 'fp32'}: {n / dt:,.0f} images/s end to end (incl. host->device copy and the "
              f"all-gather); parity vs CPU oracle on {k} images: word_acc {acc['word_acc']:.4f}, "
              f"1-N.E.D {acc['1-N.E.D']:.4f}")
        print("sample:", got[:2])
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
