"""Which HIP runtime does the process end up with: libsmhip first, then torch -- and the other way round."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = {
 "smhip_first": "import sys; sys.path.insert(0, %r); from staticmapping_amd import _capi; lib=_capi.load_library(); print('devices', lib.smhip_device_count()); import torch; print('torch sees', torch.cuda.device_count()); torch.zeros(1, device='cuda'); print('ok')" % ROOT,
 "torch_first": "import sys; sys.path.insert(0, %r); import torch; torch.zeros(1, device='cuda'); from staticmapping_amd import _capi; lib=_capi.load_library(); print('devices', lib.smhip_device_count()); print('ok')" % ROOT,
}
for k, c in code.items():
    p = subprocess.run([sys.executable, "-c", c + "; print([l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l or 'libhsa-runtime' in l][::8])"], capture_output=True, text=True)
    print(k, "rc", p.returncode, p.stdout.strip()[-600:], p.stderr.strip()[-300:])
