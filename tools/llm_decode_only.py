import sys, torch
sys.path.insert(0, ".")
from landiff_amd.config import LLMConfig
from landiff_amd.llm import LLMRunner
from landiff_amd.weights import init_state, llm_spec
dev = torch.device("cuda:0")
cfg = LLMConfig()
run = LLMRunner(init_state(llm_spec(cfg), 1, dtype=torch.bfloat16, device=dev), cfg, dev)
text = torch.randn(64, cfg.text_dim, device=dev)
run.sample(text, guidance_scale=7.5, seed=42, use_graph=False)
torch.cuda.synchronize()
