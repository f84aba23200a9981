import os, sys, time, torch
sys.path.insert(0, "/root/repo")
import fovvideovdp_amd as fv
from fovvideovdp_amd.synth import synth_video_pair
H, W, N, fps = 1080, 1920, 120, int(sys.argv[1]) if len(sys.argv) > 1 else 144
test, ref = synth_video_pair(N, H, W, device="cuda")
m = fv.fvvdp(display_name="standard_fhd")
inner = fv.fvvdp_video_source_array(test, ref, fps, display_photometry=m.display_photometry)
Lt = [inner.get_test_frame(f, torch.device("cuda")).reshape(1, 1, 1, H, W).clone() for f in range(N)]
Lr = [inner.get_reference_frame(f, torch.device("cuda")).reshape(1, 1, 1, H, W).clone() for f in range(N)]
class Resident(fv.fvvdp_video_source):
    def get_video_size(self): return (H, W, N)
    def get_frames_per_second(self): return fps
    def get_test_frame(self, f, device): return Lt[f]
    def get_reference_frame(self, f, device): return Lr[f]
def timeit(fn):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(4):
        t0 = time.perf_counter(); q = fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3, float(q[0])
print("array  %d fps: %.2f ms JOD %.6f" % ((fps,) + timeit(lambda: m.predict(test, ref, frames_per_second=fps))))
print("source %d fps: %.2f ms JOD %.6f" % ((fps,) + timeit(lambda: m.predict_video_source(Resident()))))
