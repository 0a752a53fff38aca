"""CPU timeline of the micro-batch loop of infer.synthesize (a copy of the loop with a clock after every statement)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench, fastdiff_amd
from fastdiff_amd import infer, shard, sampler, schedules
torch.manual_seed(1234)
model = fastdiff_amd.FastDiff().cuda().eval()
items = bench.config4_items()
for i in range(2):
    infer.synthesize(model, items, n_steps=6, max_batch=8, seed=i, drop_last_frame=False)
torch.cuda.synchronize()
rows = sampler.InferenceSchedule(schedules.training_hyperparams(), schedules.noise_schedule_for(6), verbose=False).rows()
lengths = [it["len"] for it in items]
hop = 256
t_max, b_max = max(lengths), 8
mel_pin = [torch.empty(b_max * 80 * t_max, dtype=torch.float32).pin_memory() for _ in range(2)]
pcm_pin = [torch.empty(b_max * t_max * hop, dtype=torch.int16).pin_memory() for _ in range(2)]
pending = None
out = {}
T0 = time.perf_counter()
def clk(label, k):
    print("k=%d %-22s %7.2f ms" % (k, label, (time.perf_counter() - T0) * 1e3))
k = 0
for batch_idx in shard.micro_batches(range(len(items)), lengths, 8):
    clk("start", k)
    mels, lens, names = infer.collate_test_batch([items[i] for i in batch_idx], False)
    clk("collate", k)
    B, _, T = mels.shape
    mel_h = mel_pin[k & 1][: B * 80 * T].view(B, 80, T)
    mel_h.copy_(mels)
    clk("copy to pinned", k)
    mels = mel_h.cuda(non_blocking=True)
    clk("h2d", k)
    with torch.no_grad():
        wav = model.sample(mels, rows, ddim=False, seed=3, lens=lens, stream_ids=list(batch_idx))
    clk("sample", k)
    pcm = model.peak_normalize_int16(wav, valid=[t * hop for t in lens])
    clk("epilogue", k)
    host = pcm_pin[k & 1][: B * T * hop].view(B, T * hop)
    host.copy_(pcm, non_blocking=True)
    clk("d2h", k)
    done = torch.cuda.Event(); done.record()
    if pending is not None:
        pending[0].synchronize()
        clk("sync prev", k)
        for b, (name, t) in enumerate(zip(pending[2], pending[3])):
            out[name] = pending[1][b, : t * hop].numpy().copy()
        clk("unpack prev", k)
    pending = (done, host, names, lens)
    k += 1
torch.cuda.synchronize(); clk("end", k)
