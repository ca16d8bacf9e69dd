import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from audiotoken_amd import weights as W
from audiotoken_amd.configs import Wav2VecBertConfig
from audiotoken_amd.encoder import Wav2VecBertEncoder
from oracle import w2vbert_ref as R
g = np.load('tests/golden/fbank_c.npz')
wave, mask = torch.from_numpy(g['wave']), torch.from_numpy(g['mask'])
enc = Wav2VecBertEncoder(Wavev := Wav2VecBertConfig(output_layer=0), device='cuda:0', quantize=False, weights=W.synth_w2vbert_weights(0, 5, False))
_, taps = enc(wave.cuda(), mask.cuda(), 10, n_layers=0, return_taps=True)
got = taps['input_features'].cpu()
ref = torch.from_numpy(g['input_features'])
def emu(mode):
    x = wave*(2**15); nf = R.num_frames(x.shape[1])
    fr = x.unfold(1,400,160)[:, :nf].clone()
    fr = fr - fr.mean(2, keepdim=True)
    prev = fr[..., :-1].clone()
    fr[...,1:] = fr[...,1:] - 0.97*prev
    fr[...,0] = fr[...,0]*(1-0.97)
    fr = fr*R.povey_window()
    k = np.arange(257); t = np.arange(400)
    ang = 2*np.pi*((np.outer(t,k))%512)/512
    M = torch.from_numpy(np.concatenate([np.cos(ang), -np.sin(ang)],1))
    spec = (fr.double() @ M)
    if mode == 'f64all':
        power = (spec[...,:257]**2 + spec[...,257:]**2)
        mel = torch.log(torch.maximum(power @ R.mel_filter_bank().double(), torch.tensor(R.MEL_FLOOR).double())).float()
    else:
        spec = spec.float(); power = spec[...,:257]**2 + spec[...,257:]**2
        mel = torch.log(torch.maximum(power @ R.mel_filter_bank(), torch.tensor(R.MEL_FLOOR)))
    fm = R.frame_mask(mask, nf).unsqueeze(-1).expand(-1,-1,80)
    masked = mel*fm; cnt = fm.sum(1,keepdim=True).clamp(min=1)
    mean = masked.sum(1,keepdim=True)/cnt
    var = (((masked-mean)**2)*fm).sum(1,keepdim=True)/cnt
    f = (mel-mean)/torch.sqrt(var+1e-7)
    f = f[:, :nf - nf%2].reshape(2,-1,160); fm2 = fm[:, :nf-nf%2].reshape(2,-1,160)
    return torch.where(fm2==0, 1.0, f), mel
e32, mel32 = emu('f32'); e64, mel64 = emu('f64all')
n = e32.shape[1]
for name, a, b in (('gpu-ref', got[:, :n], ref[:, :n]), ('gpu-emu', got[:, :n], e32), ('emu-ref', e32, ref[:, :n]), ('gpu-exact', got[:, :n], e64), ('ref-exact', ref[:, :n], e64)):
    d = (a-b).abs(); i = np.unravel_index(d.argmax().item(), d.shape)
    print(name, 'max %.3e mean %.3e at'%(d.max().item(), d.mean().item()), i, a[i].item(), b[i].item())
lm = R.log_mel(wave)
print('ref logmel at frame 117 bins 0..4', lm[1,117,:5], 'exact', mel64[1,117,:5])
