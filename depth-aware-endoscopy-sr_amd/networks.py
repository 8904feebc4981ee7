"""``define_G`` — the DepthNet branch of the reference's net factory (codes/models/networks.py:41-49).

A reference checkout switches to the HIP path by importing this ``define_G`` (or ``DepthNet``) instead
of ``models.networks.define_G``; INTEGRATION.md shows the two-line patch.
"""
from .depthnet import DepthNet


def define_G(opt):
    opt_net = opt['network_G']
    which_model = opt_net['which_model_G']
    if which_model != 'DepthNet':
        raise NotImplementedError('Generator model [{:s}] not recognized (dasr_amd builds DepthNet only)'
                                  .format(which_model))
    datalist = list(opt['datasets'].items())
    if datalist[0][0] == 'train':
        depthRangeNum = opt['datasets']['train']['depthMaskNum']
    else:
        depthRangeNum = opt['datasets']['test_1']['depthMaskNum']
    g = lambda k, d=None: opt_net[k] if (k in opt_net and opt_net[k] is not None) else d
    return DepthNet(which_ResBlk_depth=opt_net['which_ResBlk_depth'], in_nc=opt_net['in_nc'],
                    out_nc=opt_net['out_nc'], nf=opt_net['nf'], nb=opt_net['nb'], scale=opt_net['upscale'],
                    input_para=g('code_length', 10), depth_latent_ch=opt_net['depth_latent_ch'],
                    depthRangeNum=depthRangeNum, norm_type=g('norm_type', 'weight_norm'),
                    use_trainable_params=g('use_trainable_params', True), norm_gamma=g('norm_gamma', 0.1),
                    norm_beta=g('norm_beta', 0.1), ablate_depth_block=bool(g('ablate_depth_block', False)),
                    ablate_depth_matrix=bool(g('ablate_depth_matrix', False)))


X8_NETWORK_G = dict(which_model_G='DepthNet', which_ResBlk_depth=list(range(14)), in_nc=3, out_nc=3, nf=64, nb=16,
                    upscale=8, code_length=10, depth_latent_ch=256, norm_type='weight_norm',
                    use_trainable_params=True, norm_gamma=0.1, norm_beta=0.1, ablate_depth_block=False,
                    ablate_depth_matrix=False)
"""network_G of options/train/train_depthNet_SEAN_depthMask_x8.yml:47-63."""
