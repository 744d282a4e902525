"""Does a Llama-3.2-Vision model see the constant padding tiles of its pixel_values?  (CPU, tiny random model.)

The reference adds noise to the WHOLE pixel_values tensor (attack_model.py:320), padding tiles included
(llama32processor.py:344-346 fills them with zeros).  ADVX_PAD_KEEP / noise_on_padding=False keeps those tiles
exact zeros and saves three quarters of the generator work - which is the same attack only if the model cannot
see them.  It can: the vision encoder's attention mask built from `aspect_ratio_mask` blocks padding-to-padding
pairs only (transformers modeling_mllama._prepare_aspect_ratio_attention_mask: outer product of the inverted
mask), so the tokens of the real tile attend to the padding tiles' tokens, and `pixel_values.grad` on the padding
tiles is not zero.  Hence the trainers' default stays the reference's tensor (noise on every tile) and
`--no_noise_on_padding` is an explicit deviation, not an optimisation that is free."""
import pytest
import torch


def _tiny_mllama():
    from transformers import MllamaConfig, MllamaForConditionalGeneration
    from transformers.models.mllama.configuration_mllama import MllamaTextConfig, MllamaVisionConfig
    torch.manual_seed(0)
    vc = MllamaVisionConfig(hidden_size=32, num_hidden_layers=2, num_global_layers=1, attention_heads=2, intermediate_size=64,
                            image_size=28, patch_size=14, max_num_tiles=4, vision_output_dim=96, intermediate_layers_indices=[0, 1],
                            supported_aspect_ratios=[[1, 1], [1, 2], [1, 3], [1, 4], [2, 1], [2, 2], [3, 1], [4, 1]])
    tc = MllamaTextConfig(vocab_size=128, hidden_size=32, num_hidden_layers=3, cross_attention_layers=[1], num_attention_heads=2,
                          num_key_value_heads=2, intermediate_size=64, max_position_embeddings=128, pad_token_id=0,
                          bos_token_id=1, eos_token_id=2)
    model = MllamaForConditionalGeneration(MllamaConfig(vision_config=vc, text_config=tc, image_token_index=127)).eval()
    with torch.no_grad():
        for p in model.parameters():
            p.requires_grad_(False)
            if p.abs().max() == 0:      # zero-initialised gates would hide the vision path altogether
                p.normal_(0, 0.5)
    return model


@pytest.mark.timeout(300)
def test_hf_mllama_attends_to_padding_tiles():
    model = _tiny_mllama()
    B, S = 2, 10
    ids = torch.randint(3, 120, (B, S), generator=torch.Generator().manual_seed(3))
    ids[:, 0] = 127
    inputs = dict(input_ids=ids, attention_mask=torch.ones(B, S, dtype=torch.long),
                  aspect_ratio_ids=torch.ones(B, 1, dtype=torch.long))
    mask = torch.zeros(B, 1, 4, dtype=torch.long)
    mask[:, :, 0] = 1                                        # one real tile, three padding tiles
    cam = torch.zeros(B, S, 1, 4, dtype=torch.long)
    cam[:, :, 0, 0] = 1                                      # the text attends to the real tile only

    def run(pad_sigma):
        gen = torch.Generator().manual_seed(1)
        pv = torch.zeros(B, 1, 4, 3, 28, 28)
        pv[:, :, 0] = torch.randn(B, 1, 3, 28, 28, generator=gen)
        if pad_sigma:
            pv[:, :, 1:] = torch.randn(B, 1, 3, 3, 28, 28, generator=gen) * pad_sigma
        pv.requires_grad_(True)
        out = model(pixel_values=pv, aspect_ratio_mask=mask, cross_attention_mask=cam, **inputs)
        loss = out.logits.float().logsumexp(-1).mean()
        loss.backward()
        return float(loss.detach()), pv.grad[:, :, 0].clone(), float(pv.grad[:, :, 1:].abs().max())

    loss0, live0, pad_grad0 = run(0.0)
    loss1, live1, _ = run(1.0)
    assert pad_grad0 > 0.0                                   # the loss depends on the padding tiles' pixels
    assert loss1 != loss0                                    # noise there moves the loss ...
    assert float((live0 - live1).norm() / live0.norm()) > 1e-6     # ... and the gradient on the real tile
