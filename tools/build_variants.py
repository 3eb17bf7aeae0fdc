"""Builds A/B variants of liblsnet_hip.so beside the product one: python tools/build_variants.py name:-DX,-DY=3 name2:...
-> lsnet_amd/csrc/ab_<name>.so (git-ignored; travels with gpurun).  The harnesses load them through LSNET_SO, the Python
package through LSNET_HIP_SO."""
import os
import sys
import importlib.util

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location('lsn_build', os.path.join(HERE, '..', 'lsnet_amd', 'csrc', 'build.py'))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
for arg in sys.argv[1:]:
    name, _, defs = arg.partition(':')
    defines = ['LSNET_AB=1'] + [d[2:] for d in defs.split(',') if d.startswith('-D')]
    print(b.build(force=False, verbose=False, defines=defines, so=os.path.join(b.HERE, 'ab_%s.so' % name)))
