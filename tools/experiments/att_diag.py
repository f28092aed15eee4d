import torch, sys, os
sys.path.insert(0, ".")
from tests import hiputil as U
B,T,H,dh=2,135,6,80
g=torch.Generator().manual_seed(T)
qkv=(torch.randn((B*T,3*H*dh),generator=g)*0.7).to(U.DEV).to(torch.bfloat16)
o,lse=U.attention_fwd("bf16",qkv,B,T,H,dh)
q,k,v=qkv.double().reshape(B,T,3,H,dh).permute(2,0,3,1,4)
s=(q@k.transpose(-1,-2))*dh**-0.5
ref_lse=torch.logsumexp(s,-1)
print(os.environ.get("V4H_ATTN_FWD3"), "lse err", (lse.double()-ref_lse).abs().max().item())
# partial references: without d tail, without key tail
s2=(q[...,:64]@k[...,:64].transpose(-1,-2))*dh**-0.5
print("  lse err vs no-d-tail ref", (lse.double()-torch.logsumexp(s2,-1)).abs().max().item())
ref=U.ref_attention(qkv,B,T,H,dh)
print("  o err", (o.double()-ref).abs().max().item())
a=torch.softmax(s,-1)
o_nokt=(a[...,:128]@v[...,:128,:]).transpose(1,2).reshape(B*T,H*dh)
print("  o err vs no-key-tail ref", (o.double()-o_nokt).abs().max().item())
# hypothesis: keys 135..143 (clamped copies of key 134) are not masked
kx=torch.cat([k, k[..., 134:135, :].expand(-1,-1,9,-1)], dim=-2)
sx=(q@kx.transpose(-1,-2))*dh**-0.5
print("  lse err vs unmasked-clamped ref", (lse.double()-torch.logsumexp(sx,-1)).abs().max().item())
for n in (1,2,3,4,5,8):
    kx=torch.cat([k, k[..., 134:135, :].expand(-1,-1,n,-1)], dim=-2)
    print("   extra", n, (lse.double()-torch.logsumexp((q@kx.transpose(-1,-2))*dh**-0.5,-1)).abs().max().item())
