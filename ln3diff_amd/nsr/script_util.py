"""`AE` - the reconstruction-model wrapper the reference's drivers talk to (nsr/script_util.py:25-377): everything goes through
`forward(img, c, latent, behaviour, ...)` "for DDP use".  Only the decoder half is on the sampling hot path, so the behaviours
that `render_video_given_triplane` / `eval_i23d_and_export` use are implemented on the HIP decoder; the encoder behaviours
raise (the image encoder of the VAE is out of scope).

    decode_after_vae_no_render : latent dict -> + 'latent_after_vit' [B,96,128,128] (+ 'planes_channel_last' for the renderer)
    triplane_dec               : tri-planes (dict or tensor) + c [V,25] -> Triplane.forward dict (image_raw, image_depth, ...)
    decode_after_vae           : the two above in one call
    triplane_decode_grid       : tri-planes + grid_size -> {'sigma': [B,G,G,G,1], 'rgb': [B,G,G,G,3]}
    triplane_renderer          : tri-planes + coordinates / directions -> decoder output at points (forward_points)
    vit_postprocess_triplane_dec, get_rendering_kwargs
"""
import torch
import torch.nn as nn

_ENCODER_BEHAVIOURS = ('enc_dec', 'enc', 'dec', 'dec_wo_triplane', 'enc_dec_wo_triplane', 'encoder_vae')


class AE(nn.Module):
    def __init__(self, encoder, decoder, img_size, encoder_cls_token=False, decoder_cls_token=False, preprocess=None, use_clip=False,
                 dino_version='sd_dit', clip_dtype=None, no_dim_up_mlp=True, dim_up_mlp_as_func=False, uvit_skip_encoder=False,
                 confnet=None):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder
        self.img_size = img_size
        self.encoder_cls_token, self.decoder_cls_token = encoder_cls_token, decoder_cls_token
        self.use_clip, self.dino_version, self.confnet = use_clip, dino_version, confnet
        self.preprocess = preprocess
        self.dim_up_mlp = None
        self.dim_up_mlp_as_func = dim_up_mlp_as_func

    def decode_after_vae_no_render(self, ret_dict, img_size=None):
        if img_size is None:
            img_size = self.img_size
        assert self.dim_up_mlp is None
        latent = self.decoder.vit_decode_backbone(ret_dict, img_size)
        return self.decoder.vit_decode_postprocess(latent, ret_dict)

    def decode_after_vae(self, ret_dict, c, img_size=None, return_raw_only=False, **kwargs):
        ret_dict = self.decode_after_vae_no_render(ret_dict, img_size)
        return self.decoder.triplane_decode(ret_dict, c, return_raw_only=return_raw_only, **kwargs)

    @torch.no_grad()
    def forward(self, img=None, c=None, latent=None, behaviour='enc_dec', coordinates=None, directions=None, return_raw_only=False,
                *args, **kwargs):
        if behaviour in _ENCODER_BEHAVIOURS:
            raise NotImplementedError(f"AE behaviour '{behaviour}' needs the VAE encoder, which is outside the sampling hot path")
        if behaviour == 'decode_after_vae_no_render':
            return self.decode_after_vae_no_render(latent, self.img_size)
        if behaviour == 'decode_after_vae':
            return self.decode_after_vae(latent, c, self.img_size, return_raw_only, **kwargs)
        if behaviour == 'triplane_dec':
            assert latent is not None
            return self.decoder.triplane_decode(latent, c, return_raw_only=return_raw_only, **kwargs)
        if behaviour == 'triplane_decode_grid':
            assert latent is not None
            return self.decoder.triplane_decode_grid(latent, **kwargs)
        if behaviour == 'vit_postprocess_triplane_dec':
            assert latent is not None
            return self.decoder.triplane_decode(self.decoder.vit_decode_postprocess(latent, {}), c)
        if behaviour == 'triplane_renderer':
            assert latent is not None
            return self.decoder.triplane_renderer(latent, coordinates, directions)
        if behaviour == 'get_rendering_kwargs':
            return self.decoder.triplane_decoder.rendering_kwargs
        raise ValueError(f"unknown AE behaviour '{behaviour}'")


class AE_with_Diffusion(nn.Module):          # nsr/script_util.py:386-410 (container used by the joint trainers)
    def __init__(self, auto_encoder, denoise_model):
        super().__init__()
        self.auto_encoder = auto_encoder
        self.denoise_model = denoise_model

    def forward(self, img, c, behaviour='enc_dec', latent=None, *args, **kwargs):
        return self.auto_encoder(img, c, behaviour=behaviour, latent=latent, *args, **kwargs)
