python -c "
from nolzss_amd import _noLZSS as n
print('ours', n.device_count(), n.count_factors(b'abracadabra'))
import torch
print('torch', torch.cuda.is_available(), torch.zeros(3, device='cuda').sum().item())
" 2>&1 | tail -2
python -c "
import noLZSS
print(noLZSS.factorize(b'abcabcabc'))
import torch
print('torch after noLZSS', torch.cuda.is_available())
" 2>&1 | tail -2
