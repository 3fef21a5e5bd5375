import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ser_amd
from transformers import Wav2Vec2Config, XLMRobertaConfig
from ser_amd.models import AudioEncoder, TextEncoder
from ser_amd.system import SERSystem, TrainStepper
dev = torch.device("cuda:0")
LD = float(os.environ.get("LD", "0.3")); MP = float(os.environ.get("MP", "0.3")); DROP = os.environ.get("DROP", "1") == "1"
NOISE = os.environ.get("NOISE", "1") == "1"
def build():
    torch.manual_seed(0)
    wc = Wav2Vec2Config(hidden_size=128, num_hidden_layers=3, num_attention_heads=2, intermediate_size=256, conv_dim=[64] * 7,
                        num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4, layerdrop=LD, mask_time_prob=MP,
                        mask_time_length=2, mask_time_min_masks=2)
    xc = XLMRobertaConfig(vocab_size=1000, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                          max_position_embeddings=66, type_vocab_size=1, layer_norm_eps=1e-5, pad_token_id=1, bos_token_id=0, eos_token_id=2)
    ae = AudioEncoder(hf_config=wc, adapter_dim=32, freeze_base=False, use_quality_gates=False, use_audio_conditioning=False)
    te = TextEncoder(hf_config=xc, adapter_dim=32, freeze_base=False)
    sysm = SERSystem(ae, te, num_labels=4, shared_dim=64, num_heads=2, proj_dim=64, num_layers=3, base_dim=64).to(dev)
    sysm.train()
    sysm.train_dropout = DROP
    if NOISE:
        for m in (sysm.audio_encoder, sysm.text_encoder):
            m.encoder_train_noise, m.noise_seed = True, 5
    return sysm
g = torch.Generator().manual_seed(11)
B, T, S = 3, 4000, 9
batches = []
for _ in range(3):
    ids = torch.randint(4, 1000, (B, S), generator=g)
    ids[:, 0], ids[:, -1] = 0, 2
    batches.append([0.1 * torch.randn(B, T, generator=g).to(dev), ids.to(dev), torch.ones(B, S).to(dev), torch.randint(0, 4, (B,), generator=g).to(dev)])
runs = {}
for mode in ("eager", "graph"):
    sysm = build()
    opt = sysm.make_optimizer(lr=1e-3)
    st = TrainStepper(sysm, opt, use_graph=(mode == "graph"))
    rec = []
    for b in batches:
        loss = st.step(*b).clone()
        torch.cuda.synchronize()
        rec.append((loss.item(), st.logits.detach().cpu().clone(), {n: p.detach().cpu().clone() for n, p in sysm.named_parameters()},
                    {n: (None if p.grad is None else p.grad.detach().cpu().clone()) for n, p in sysm.named_parameters()},
                    sorted(sysm.audio_encoder._noise.skip) if NOISE else []))
    runs[mode] = rec
for i, (e, gq) in enumerate(zip(runs["eager"], runs["graph"])):
    print(f"step {i}: loss {e[0]!r} vs {gq[0]!r}  logits max diff {(e[1]-gq[1]).abs().max().item():.3e} skip {e[4]} {gq[4]}")
    bad = [(n, (v - gq[2][n]).abs().max().item()) for n, v in e[2].items() if not torch.equal(v, gq[2][n])]
    print(f"   params differing after the step: {len(bad)} of {len(e[2])}", bad[:6])
    badg = []
    for n, v in e[3].items():
        w = gq[3][n]
        if v is None:
            if w is not None and w.abs().max().item() != 0: badg.append((n, "eager None, graph nonzero"))
        elif w is None: badg.append((n, "graph None"))
        elif not torch.equal(v, w): badg.append((n, (v - w).abs().max().item(), v.abs().max().item()))
    print(f"   grads differing: {len(badg)}", badg[:8])
