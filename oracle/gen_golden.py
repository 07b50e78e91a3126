"""Generate tests/golden/*.npz by running the REFERENCE's own code in the build container.

Run here only (``python oracle/gen_golden.py``): it imports ``/root/reference/training/model.py``
(quantizer + encoder classes) and loads the shipped checkpoint with ``weights_only=True``.
The reference never travels: only the inputs/outputs written below are committed.
``seaborn`` (absent here, used by the reference only for one heat-map at training/model.py:733)
is replaced by an empty module.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def main():
    sys.dont_write_bytecode = True
    sys.modules.setdefault("seaborn", types.ModuleType("seaborn"))
    sys.path.insert(0, os.path.join(REF, "training"))
    import model as ref  # noqa: E402  (reference training/model.py)

    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)  # fixed reduction order for the golden distances
    sd = torch.load(os.path.join(REF, "training", "vq_vae_model.pth"), map_location="cpu", weights_only=True)
    m = ref.VQVAE(30, ref.VQ_EMBEDDING_DIM, ref.VQ_NUM_EMBEDDINGS, ref.VQ_COMMITMENT_COST, ref.VQ_DECAY)
    m.load_state_dict(sd)
    m.eval()

    # (1) shipped weights needed on the hot path (encoder + codebook + EMA buffers) and by VQVAE.forward (decoder)
    keep = {k: v.numpy() for k, v in sd.items() if k.startswith(("encoder.", "vq_layer.", "decoder."))}
    np.savez_compressed(os.path.join(OUT, "vq_vae_shipped.npz"), **keep)
    cb = sd["vq_layer.embedding.weight"].numpy()
    zero_rows = np.nonzero((cb == 0).all(1))[0]
    print("codebook", cb.shape, "zero rows", len(zero_rows), "first", zero_rows[:5],
          "unique rows", len(np.unique(cb, axis=0)), "max |e|^2", (cb.astype(np.float64) ** 2).sum(1).max())

    out = {}
    with torch.no_grad():
        # (2) VQVAE.encode on seeded epochs of the three lengths the survey verified
        g = torch.Generator().manual_seed(0)
        for name, shape in (("S", (2, 30, 1500)), ("R", (2, 30, 1001)), ("P", (4, 30, 39))):
            x = torch.randn(*shape, generator=g)
            z = m.encoder(x)
            tok = m.encode(x)
            out[f"x_{name}"] = x.numpy()
            out[f"z_{name}"] = z.numpy()
            out[f"tokens_{name}"] = tok.numpy()
        # (3) VQVAE.forward / VectorQuantizerEMA.forward (eval mode) on L=39 (L % 4 == 3 required)
        x = torch.from_numpy(out["x_P"])
        vq_loss, x_recon, perplexity, indices = m(x)
        out["fwd_vq_loss"] = np.float32(vq_loss.item())
        out["fwd_perplexity"] = np.float32(perplexity.item())
        out["fwd_indices"] = indices.numpy()
        out["fwd_x_recon"] = x_recon.numpy()
        zp = m.encoder(x).permute(0, 2, 1).contiguous()
        loss2, quant2, perp2, idx2 = m.vq_layer(zp)
        out["vqfwd_in"] = zp.numpy()
        out["vqfwd_loss"] = np.float32(loss2.item())
        out["vqfwd_quantized"] = quant2.numpy()
        out["vqfwd_perplexity"] = np.float32(perp2.item())
        out["vqfwd_indices"] = idx2.numpy()
        # (4) get_code_indices on rows drawn near live codes, plus exact-tie rows
        live = np.setdiff1d(np.arange(cb.shape[0]), zero_rows)
        rs = np.random.RandomState(1234)
        pick = rs.choice(live, size=1024)
        flat = cb[pick] + rs.normal(0, 0.05, size=(1024, cb.shape[1])).astype(np.float32)
        flat[:8] = 0.0                        # ties among the all-zero duplicate codes
        flat[8:16] = cb[pick[8:16]]           # exactly a codebook row
        flat = flat.astype(np.float32)
        idx = m.vq_layer.get_code_indices(torch.from_numpy(flat))
        out["gci_in"] = flat
        out["gci_idx"] = idx.numpy()
        out["gci_pick"] = pick.astype(np.int64)
        # (5) training-mode EMA update of the quantizer (one step) from the shipped state
        vq = ref.VectorQuantizerEMA(512, 256, 0.25, 0.99)
        vq.load_state_dict({k[len("vq_layer."):]: v.clone() for k, v in sd.items() if k.startswith("vq_layer.")})
        vq.train()
        inp = torch.from_numpy(flat[:512]).view(8, 64, 256)
        loss_t, q_t, perp_t, idx_t = vq(inp)
        out["ema_in"] = inp.numpy()
        out["ema_idx"] = idx_t.numpy()
        out["ema_cluster_size"] = vq.ema_cluster_size.numpy()
        out["ema_w"] = vq.ema_w.numpy()
        out["ema_weight"] = vq.embedding.weight.detach().numpy()
        out["ema_loss"] = np.float32(loss_t.item())
        out["ema_perplexity"] = np.float32(perp_t.item())
    np.savez_compressed(os.path.join(OUT, "vq_golden.npz"), **out)
    print("forward: loss", out["fwd_vq_loss"], "perplexity", out["fwd_perplexity"],
          "tokens_S[0,:8]", out["tokens_S"][0, :8], "gci first 16", out["gci_idx"][:16])

    # (6) band-power -> label known-answer test from the reference's shipped arrays
    X = np.load(os.path.join(REF, "X_psd_multi.npy"))
    y = np.load(os.path.join(REF, "y_label_multi.npy"))
    y_var = np.load(os.path.join(REF, "unsupervised", "y_label_multi.npy"))
    freqs = np.fft.rfftfreq(512, 1.0 / 250.0)
    freqs = freqs[(freqs >= 4.0) & (freqs <= 30.0)]
    assert X.shape[-1] == len(freqs), (X.shape, len(freqs))
    np.savez_compressed(os.path.join(OUT, "bandpower_kat.npz"), log_psd=X, labels=y, labels_variant=y_var, freqs=freqs)
    print("KAT", X.shape, y.shape, np.bincount(y), np.bincount(y_var))


if __name__ == "__main__":
    main()
