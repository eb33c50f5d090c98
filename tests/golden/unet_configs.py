"""U-Net configurations and the synthetic-weight recipe shared by tests/golden/make_golden_unet.py and the tests (data, no reference code)."""
import torch

from ln3diff_amd.synth import synth_state_dict

CONFIGS = {
    # tiny spatial-transformer U-Net: scale-shift ResBlocks, transformers at both levels (16 and 64 tokens), down + up sample
    'tiny_st': dict(image_size=16, in_channels=4, model_channels=128, out_channels=4, num_res_blocks=1, attention_resolutions=[2, 1],
                    channel_mult=(1, 2), num_heads=4, use_spatial_transformer=True, transformer_depth=1, context_dim=768,
                    use_scale_shift_norm=True, roll_out=False),
    # AttentionBlock variant (no context), plain `h + emb` ResBlocks, rolled-out tri-plane input [B, 3C, S, S] -> [B, C, S, 3S]
    'tiny_attn': dict(image_size=16, in_channels=4, model_channels=128, out_channels=4, num_res_blocks=1, attention_resolutions=[2, 1],
                      channel_mult=(1, 2), num_heads=4, use_spatial_transformer=False, context_dim=None, use_scale_shift_norm=False,
                      roll_out=True),
    # shell_scripts/final_release/inference/sample_shapenet_*_t23d.sh: --num_channels 320 --num_res_blocks 2 --num_heads 8
    # --attention_resolutions 4,2,1 --use_spatial_transformer True --context_dim 768, 12-channel 32 x 32 latent (827 M parameters)
    'shapenet': dict(image_size=32, in_channels=12, model_channels=320, out_channels=12, num_res_blocks=2, attention_resolutions=[8, 16, 32],
                     channel_mult=(1, 2, 4, 4), num_heads=8, use_spatial_transformer=True, transformer_depth=1, context_dim=768,
                     use_scale_shift_norm=True, roll_out=False),
}



def synth_unet_sd(shapes, seed):
    """(name, shape, seed) weights; mixing_logit gets values that make the mixture non-trivial (its init is a constant -6)."""
    sd = synth_state_dict(shapes, seed)
    if 'mixing_logit' in sd:
        sd['mixing_logit'] = torch.linspace(-1.5, 1.5, sd['mixing_logit'].numel()).reshape(sd['mixing_logit'].shape)
    return sd


