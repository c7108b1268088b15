"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads and exports every symbol
declared in include/k4nerf.h, the Python mirror keeps the reference's names, and there is no CPU fallback."""
import os
import re

import numpy as np
import pytest
import torch

import nerf4k_amd  # noqa: F401
from nerf4k_amd import _native as N, scene
from nerf4k_amd.lib import dvgo, dmpigo, grid, render_utils_cuda, utils
from helpers import GOLDEN, load_march_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'k4nerf.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(k4_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    lib = N.lib()
    syms = _declared_symbols()
    assert len(syms) >= 17
    for s in syms:
        assert hasattr(lib, s), f'{s} declared in include/k4nerf.h but not exported by lib4k_hip.so'
    assert lib.k4_abi_version() == N.K4_ABI_VERSION


def test_render_utils_shim_has_the_13_reference_names():
    # /root/reference/lib/cuda/render_utils.cpp:170-184
    names = ['infer_t_minmax', 'infer_n_samples', 'infer_ray_start_dir', 'sample_pts_on_rays',
             'sample_ndc_pts_on_rays', 'sample_bg_pts_on_rays', 'maskcache_lookup', 'raw2alpha',
             'raw2alpha_backward', 'raw2alpha_nonuni', 'raw2alpha_nonuni_backward', 'alpha2weight',
             'alpha2weight_backward']
    for n in names:
        assert callable(getattr(render_utils_cuda, n))
    assert dvgo.render_utils_cuda is render_utils_cuda and hasattr(dvgo, 'Raw2Alpha') and hasattr(dvgo, 'Alphas2Weights')


@pytest.mark.parametrize('name', ['march_mpi_base', 'march_mpi_pe', 'march_dvgo_base', 'march_dvgo_nodirect',
                                  'march_dvgo_coarse'])
def test_checkpoint_contract(name):
    """load_model semantics (lib/utils.py:62-66): model_class(**model_kwargs) + strict load_state_dict of the
    reference's key names; get_kwargs() round-trips."""
    g = load_march_golden(name)
    model = utils.model_from_checkpoint_dict(g)
    assert set(model.state_dict().keys()) == set(g['model_state_dict'].keys())
    kw = model.get_kwargs()
    for k in ('xyz_min', 'xyz_max', 'num_voxels', 'mask_cache_world_size', 'fast_color_thres',
              'rgbnet_dim', 'rgbnet_depth', 'rgbnet_width', 'viewbase_pe', 'mode_type', 'act_type', 'dim_rend'):
        assert k in kw
    cls = type(model)
    model2 = cls(**kw)
    model2.load_state_dict(model.state_dict())


def test_no_cpu_fallback():
    g = load_march_golden('march_mpi_base')
    model = utils.model_from_checkpoint_dict(g)
    r = g['rays']
    with pytest.raises(N.K4Error):
        model(r['rays_o'], r['rays_d'], r['viewdirs'], **g['render_kwargs'])
    with pytest.raises(N.K4Error):
        render_utils_cuda.raw2alpha(torch.zeros(4), 0, 1.0)
    # no PyTorch back doors either: an rgbnet shape the HIP kernels do not cover, SFTNet's fea / dswise variants and CPU inputs raise
    from nerf4k_amd.lib import sr_esrnet
    import torch.nn as nn
    # (CPU input: raises whatever the shape; the shape checks themselves are exercised on the GPU in
    #  tests/test_train_ops_gpu.py::test_rgbnet_layer_by_layer_path_and_rejected_shapes)
    model.rgbnet = nn.Sequential(nn.Linear(15, 48), nn.Tanh(), nn.Linear(48, 3))           # an activation the kernels do not have
    with pytest.raises(N.K4Error):
        model._k4_rgbnet_sigmoid(torch.zeros(4, 15))
    net = sr_esrnet.SFTNet(3, scale=4, num_block=1)
    with pytest.raises(N.K4Error):
        net(torch.zeros(1, 3, 8, 8), torch.zeros(1, 1, 8, 8))
    # the decoder's sub-blocks are parameter containers: calling one directly must not evaluate it with PyTorch
    blk = net.body[0]
    for call in (lambda: blk.sft0(torch.zeros(1, 64, 4, 4), torch.zeros(1, 32, 4, 4)),
                 lambda: blk.rdb1((torch.zeros(1, 64, 4, 4), torch.zeros(1, 32, 4, 4))),
                 lambda: blk((torch.zeros(1, 64, 4, 4), torch.zeros(1, 32, 4, 4)))):
        with pytest.raises(N.K4Error):
            call()


def test_no_pytorch_fallback_in_the_product_source():
    """Static: the product package holds no path that evaluates the rgbnet or the decoder with PyTorch modules."""
    import ast, glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in glob.glob(os.path.join(root, '4k-nerf_amd', '**', '*.py'), recursive=True):
        rel = os.path.relpath(f, root)
        for node in ast.walk(ast.parse(open(f).read())):          # code, not prose: docstrings may cite the reference's expressions
            if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute):
                assert node.func.attr not in ('rgbnet', '_forward_torch'), (rel, node.lineno)
            if isinstance(node, ast.FunctionDef):
                assert node.name != '_forward_torch', (rel, node.lineno)
            if isinstance(node, ast.Constant) and isinstance(node.value, str):
                assert node.value not in ('K4_SR_TRAIN', 'K4_RGBNET'), (rel, node.lineno)
            # an nn.Module.forward that evaluates layers with PyTorch: F.<op>(...) or self.<sub-module>(...) inside a method named forward
            if isinstance(node, ast.FunctionDef) and node.name == 'forward' and node.args.args and node.args.args[0].arg == 'self':
                for sub in ast.walk(node):
                    if not (isinstance(sub, ast.Call) and isinstance(sub.func, ast.Attribute) and isinstance(sub.func.value, ast.Name)):
                        continue
                    assert sub.func.value.id != 'F', (rel, sub.lineno, 'torch.nn.functional call inside a Module.forward')
                    if sub.func.value.id == 'self':
                        assert not re.match(r'(conv|sft|rdb|lrelu|SFT_|rgbnet|body|CondNet)', sub.func.attr), \
                            (rel, sub.lineno, f'self.{sub.func.attr}(...) evaluates a PyTorch sub-module inside a Module.forward')


def test_at_most_fifteen_environment_switches():
    """Experiment knobs are module attributes (tests monkeypatch them); what the package and the library read from the environment
    stays a short, documented list (README.md)."""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set()
    for f in glob.glob(os.path.join(root, '4k-nerf_amd', '**', '*.py'), recursive=True):
        names |= set(re.findall(r"environ(?:\.get)?[\(\[]\s*'(K4_[A-Z0-9_]+)'", open(f).read()))
    for f in glob.glob(os.path.join(root, '4k-nerf_amd', 'csrc', '*')):
        names |= set(re.findall(r'env_int\("(K4_[A-Z0-9_]+)"', open(f).read())) | set(re.findall(r'getenv\("(K4_[A-Z0-9_]+)"', open(f).read()))
    assert 0 < len(names) <= 15, sorted(names)
    readme = open(os.path.join(root, 'README.md')).read()
    for n in names:
        assert n in readme, f'{n} is read from the environment but not documented in README.md'


def test_oracle_and_reference_stay_behind_the_test_boundary():
    """Static check of the scope contract: the product package and tools/ never import oracle/ (test infrastructure), and nothing
    that runs on the GPU box (product, tools, bench.py, __graft_entry__.py, tests) opens /root/reference except the two
    build-container scripts under oracle/ that make the goldens / the reference-compiled checker."""
    import glob, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    imp = re.compile(r'^\s*(from\s+oracle\b|import\s+oracle\b)', re.M)
    product = glob.glob(os.path.join(root, '4k-nerf_amd', '**', '*.py'), recursive=True) + [os.path.join(root, 'nerf4k_amd.py')]
    tools = glob.glob(os.path.join(root, 'tools', '**', '*.py'), recursive=True)
    assert product and tools
    for f in product + tools:
        assert not imp.search(open(f).read()), f'{os.path.relpath(f, root)} imports oracle/'
    # bench.py: the oracle only inside the legs that check or time it (never at module level)
    bench = open(os.path.join(root, 'bench.py')).read()
    assert not re.search(r'^(from\s+oracle|import\s+oracle)', bench, re.M)
    runs_on_gpu_box = product + tools + [os.path.join(root, 'bench.py'), os.path.join(root, '__graft_entry__.py')] + \
        glob.glob(os.path.join(root, 'tests', '**', '*.py'), recursive=True)
    import ast
    for f in runs_on_gpu_box:
        for node in ast.walk(ast.parse(open(f).read())):        # a PATH literal (citations in docstrings / comments are prose, not paths)
            if isinstance(node, ast.Constant) and isinstance(node.value, str) and re.fullmatch(r'/root/reference[\w/.\-]*', node.value.strip()):
                raise AssertionError(f'{os.path.relpath(f, root)}:{node.lineno} holds the path {node.value!r}')


def test_rays_match_reference_golden():
    z = np.load(os.path.join(GOLDEN, 'rays_views.npz'))
    H, W = int(z['H']), int(z['W'])
    for tag, ndc in (('ndc', True), ('persp', False)):
        ro, rd, vd = dvgo.get_rays_of_a_view(H, W, z[f'{tag}/K'], torch.from_numpy(z[f'{tag}/c2w']), ndc,
                                             inverse_y=False, flip_x=False, flip_y=False)
        for a, k in ((ro, 'rays_o'), (rd, 'rays_d'), (vd, 'viewdirs')):
            assert torch.allclose(a, torch.from_numpy(z[f'{tag}/{k}']), rtol=0, atol=1e-6), (tag, k)


def test_llff_scene_matches_reference_config():
    ck = scene.make_llff_checkpoint(num_voxels=48 * 48 * 32, mpi_depth=32)
    m = utils.model_from_checkpoint_dict(ck)
    assert list(m.world_size) == list(ck['model_state_dict']['density.grid'].shape[2:])
    assert m.act_shift.grid.shape == (1, 1, 1, 1, 32)
    # the per-plane bias our module builds equals the one in the checkpoint (lib/dmpigo.py:53-58)
    m2 = dmpigo.DirectMPIGO(**ck['model_kwargs'])
    assert torch.allclose(m2.act_shift.grid, ck['model_state_dict']['act_shift.grid'])


def test_one_abi_number_everywhere():
    """The header, the loader, the library and the two documents that quote the ABI version agree (round-4 verdict: INTEGRATION said 6,
    DESIGN said 7, the header 8)."""
    hdr = open(os.path.join(ROOT, 'include', 'k4nerf.h')).read()
    v = int(re.search(r'#define\s+K4_ABI_VERSION\s+(\d+)', hdr).group(1))
    assert v == N.K4_ABI_VERSION == N.lib().k4_abi_version()
    for doc in ('INTEGRATION.md', 'DESIGN.md'):
        quoted = [int(x) for x in re.findall(r'ABI version (\d+)', open(os.path.join(ROOT, doc)).read())]
        assert quoted and all(q == v for q in quoted), (doc, quoted, v)


def test_run_py_import_line_resolves():
    """`run.py:11` of the reference imports `lib.img_encoder`, which the reference tree does not ship: the package
    provides the (empty) module, and every in-scope name of that import line and of run_sr.py:13-17."""
    import importlib
    for name in ('img_encoder', 'utils', 'dvgo', 'dmpigo', 'sr_esrnet', 'masked_adam', 'grid'):
        importlib.import_module('nerf4k_amd.lib.' + name)
    import inspect
    sig = inspect.signature(dvgo.mimg_patch_indices_generator)
    assert list(sig.parameters) == ['imsz', 'num_im', 'BS', 'sz_patch', 'sr_ratio']       # lib/dvgo.py:850 (run.py:429 passes 3: a reference defect)
