"""Time of the device mel front-end for the benchmark batch (8 x 10.03 s)."""
import sys, torch
sys.path.insert(0, '/root/repo')
import fastdiff_amd
m = fastdiff_amd.FastDiff().cuda().eval()
wav = (torch.rand(8, 864 * 256 - 256) * 2 - 1).cuda() * 0.3
for _ in range(3): mel = m.mel_spectrogram(wav)
torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): mel = m.mel_spectrogram(wav)
e.record(); torch.cuda.synchronize()
print("mel front-end, B=8 x %d frames: %.1f us per call" % (mel.shape[-1], s.elapsed_time(e) * 100))
