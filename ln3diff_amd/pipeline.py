"""End-to-end text->3D sampling pipeline on the HIP path (restates the reference drivers
DiffusionEngineLSGM.sample/eval_cldm (nsr/lsgm/sgm_DiffusionEngine.py:386-480) and
render_video_given_triplane (nsr/train_util_diffusion.py:177-300)): noise -> EulerEDM x num_steps with CFG
-> latent * triplane_scaling_divider -> VAE decode -> V views per sample through the fused ray-marcher."""
import torch

from .sgm.sampling import DiscreteDenoiser, EulerEDMSampler, VanillaCFG

TRIPLANE_SCALING_DIVIDER = 0.96806      # nsr/train_util_diffusion.py:188


class T23DPipeline:
    def __init__(self, dit, decoder, num_steps=250, cfg_scale=6.5, conditioner=None):
        self.dit, self.decoder, self.conditioner = dit, decoder, conditioner
        self.sampler = EulerEDMSampler(num_steps=num_steps, guider=VanillaCFG(cfg_scale))
        self.denoiser = DiscreteDenoiser()

    @torch.no_grad()
    def encode_prompts(self, captions, uc_captions=None):
        """GeneralConditioner semantics for the T23D config (sgm/modules/encoders/modules.py:80-191,
        sgm/configs/txt2img-clipl-compat.yaml): one FrozenCLIPEmbedder on key 'caption' ->
        cond = {'crossattn': last_hidden_state, 'vector': pooled}; uc = the same for the legacy ucg value "" (or the ids
        passed in uc_captions).  captions: list of strings (needs the BPE vocabulary) or int token ids [B, 77]."""
        assert self.conditioner is not None, "construct the pipeline with conditioner=FrozenCLIPEmbedder(...)"
        z, pooled = self.conditioner(captions)
        B = z.shape[0]
        if uc_captions is None:
            uc_captions = [""] * B
        zu, pu = self.conditioner(uc_captions)
        return {'crossattn': z, 'vector': pooled}, {'crossattn': zu.clone(), 'vector': pu.clone()}

    @torch.no_grad()
    def sample_latent(self, z, cond, uc):
        return self.sampler(self.denoiser, self.dit, z, cond, uc)

    @torch.no_grad()
    def decode(self, latent, want_nchw=False):
        lat = latent * TRIPLANE_SCALING_DIVIDER
        tok = self.decoder.vit_decode_backbone({'latent_normalized_2Ddiffusion': lat}, 128)
        return self.decoder.vit_decode_postprocess(tok, {}, want_nchw=want_nchw)

    @torch.no_grad()
    def render(self, dec_out, cams, res, jitter=None, u_fine=None):
        """cams [V,25] rendered for EVERY sample: returns images [B, V, 3, res, res]."""
        pcl = dec_out['planes_channel_last']
        B, V = pcl.shape[0], cams.shape[0]
        c = cams.repeat(B, 1)
        idx = torch.arange(B, device=pcl.device, dtype=torch.int32).repeat_interleave(V)
        out = self.decoder.triplane_decoder(c=c, planes_channel_last=pcl, plane_index=idx,
                                            neural_rendering_resolution=res, jitter=jitter, u_fine=u_fine)
        return {k: (v.view(B, V, *v.shape[1:]) if torch.is_tensor(v) else v) for k, v in out.items()
                if k in ('image_raw', 'image_depth', 'weights_samples', 'image_mask')}

    @torch.no_grad()
    def __call__(self, z, cond, uc, cams, res):
        latent = self.sample_latent(z, cond, uc)
        dec = self.decode(latent)
        img = self.render(dec, cams, res)
        return latent, img
