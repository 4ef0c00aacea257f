"""Device-memory level after forward / backward / optimizer step of BERT-large (leak and peak check): python tools/mem_probe.py [batch]"""
import torch, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_distributed_training_amd.engine import Engine
from pytorch_distributed_training_amd.models.bert import build_model
from pytorch_distributed_training_amd.optim import AdamW
from pytorch_distributed_training_amd.data.synthetic import fixed_batch
eng = Engine(fp16=True)
model = build_model("bert-large-cased")
opt = AdamW(params=model.parameters(), lr=1e-5)
model, opt = eng.prepare(model, opt)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
batch = fixed_batch(B, 128, model.config.vocab_size, "cuda")
model.train()
G = 2 ** 30
def m(tag):
    torch.cuda.synchronize()
    print(f"{tag}: alloc {torch.cuda.memory_allocated()/G:.1f} GB  peak {torch.cuda.max_memory_allocated()/G:.1f} GB  reserved {torch.cuda.memory_reserved()/G:.1f} GB", flush=True)
m("base")
for it in range(3):
    torch.cuda.reset_peak_memory_stats()
    out = model(**batch)
    m(f"it{it} after fwd")
    eng.backward(out.loss)
    m(f"it{it} after bwd")
    opt.step(); opt.zero_grad()
    m(f"it{it} after step")
