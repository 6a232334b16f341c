"""The scoring contract of `hap` (ha/score.py:57-83) on top of haloop_amd.attention.GPT.

`hap` reads sentences, encodes them with sentencepiece, and prints per sentence
``loss_per_token<TAB>num_tokens<TAB>len(tokens)``.  Tokenisation is outside the hot path; this module
takes the token id lists and reproduces everything after it: right-padding with 0, truncation to
``block_size``, the ``[eos] + completions[:-1]`` inputs, ``forward_all(reduction='none')``, the
per-sentence sum over its row and the division by ``min(block_size, len(tokens))``.
"""
import torch

EOS = 50256          # ha/score.py:54


@torch.inference_mode()
def score_token_batches(model, completion_tokens, eos=EOS):
    """completion_tokens: list of token-id lists.  -> list of (loss_per_token, num_tokens, len(tokens))."""
    device = next(model.parameters()).device
    block = model.config.block_size
    completions = torch.nn.utils.rnn.pad_sequence([torch.LongTensor(p) for p in completion_tokens], batch_first=True,
                                                  padding_value=0).to(device)
    if completions.size(-1) >= block:
        completions = completions[:, :block].contiguous()
    prompts = torch.full((len(completions), 1), eos, dtype=torch.long, device=device)
    input_ids = torch.cat([prompts, completions[..., :-1]], dim=-1)[:, :block]
    losses = model.forward_all(input_ids=input_ids, target_ids=completions, reduction='none').view(-1, input_ids.shape[-1])
    sums = losses.sum(-1).tolist()
    out = []
    for s, tokens in zip(sums, completion_tokens):
        n = min(block, len(tokens))
        out.append((s / n, n, len(tokens)))
    return out


def format_lines(results):
    """The lines `hap` prints (non-verbose): '%.3f<TAB>num_tokens<TAB>len'."""
    return [f'{lpt:0.3f}\t{n}\t{ln}' for lpt, n, ln in results]
