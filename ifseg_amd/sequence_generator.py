"""Fixed-length decode of the segmentation tokens (mirror of models/sequence_generator.py:210-585, BASELINE config 5).

The reference's task builds `SequenceGenerator([model], eval_args {"beam":5,"max_len":1024,"min_len":1024})`
(tasks/mm_tasks/segmentation.py:166-174) but never calls it -- validation goes through the criterion -- and the class
cannot run on its own model: `_generate` reshapes the step log-probabilities with a hard-coded 151 classes (:416) and the
surrogate decoder's incremental step breaks on the attention bias (unify_multihead_attention.py:465, size 130 vs 65 on
the 8x8 fixture; reproduced in this container).  There is therefore no reference output to pin: PARITY UNPINNED.

What its loop WOULD compute is well defined, and that is what is built here:
  * `max_len = min_len = prev_output_tokens.size(1) - 1` (:229), tokens[:, 0] = bos (:303), one token per step, EOS is
    never finalised (`eos_mask` is all False, :431), the search stops at `max_len` (:465) and the best beam's tokens
    without bos are returned ([B, max_len], :561-568);
  * the SURROGATE decoder (decoder_module.py:486-677) feeds the encoder's patch outputs as decoder inputs, so the
    distribution of step t does not depend on the tokens generated before it: "incremental decoding" of all P + 1
    positions is exactly ONE causal teacher-forced pass of the HIP engine (logits [B, P+1, nseg]);
  * fairseq BeamSearch.step (search.py:102-145): candidates = cumulative score + log-prob of the step, top 2*beam over
    beam x vocabulary, the first `beam` active ones survive.  For per-step independent distributions this is the exact
    k-best search; the best beam is the per-position argmax.
Step t scores position t of the causal pass (position 0 = bos sees nothing but itself), as the reference's step t would.
"""
import torch


class SequenceGenerator:
    def __init__(self, models, tgt_dict=None, beam_size=1, max_len=None, min_len=1, temperature=1.0, **unused):
        self.models = models if isinstance(models, (list, tuple)) else [models]
        self.tgt_dict = tgt_dict
        self.beam_size = int(beam_size)
        self.temperature = float(temperature)
        assert self.temperature > 0, "--temperature must be greater than 0"

    @torch.no_grad()
    def generate(self, models, sample, **kwargs):
        return self._generate(models or self.models, sample, **kwargs)

    @torch.no_grad()
    def step_log_probs(self, model, net_input):
        """[B, T_dec, nseg] log-probabilities of every decode step from one causal pass (T_dec = P + 1)"""
        was_training = model.training
        model.eval()
        try:
            logits, _ = model(**net_input, full_context_alignment=False)
        finally:
            if was_training:
                model.train()
        return torch.log_softmax(logits.float() / self.temperature, dim=-1)

    @torch.no_grad()
    def _generate(self, models, sample, return_all_beams=False):
        net_input = sample["net_input"]
        max_len = net_input["prev_output_tokens"].size(1) - 1                     # :229
        model = models[0]
        lprobs = self.step_log_probs(model, {k: v for k, v in net_input.items()
                                             if k != "prev_output_tokens"} | {"prev_output_tokens": net_input["prev_output_tokens"][:, :1]})
        if max_len > lprobs.size(1):
            raise ValueError("prev_output_tokens asks for %d steps, the decoder has %d positions" % (max_len, lprobs.size(1)))
        tokens, scores = beam_search_independent(lprobs[:, :max_len], self.beam_size)
        best = scores[:, :, -1].argmax(1)                                         # :561-566
        pred = tokens[torch.arange(tokens.size(0), device=tokens.device), best]
        return (pred, tokens, scores) if return_all_beams else pred


def beam_search_independent(lprobs, beam_size):
    """fairseq BeamSearch over per-step independent distributions.  lprobs [B, T, V] -> (tokens [B, beam, T] int64,
    cumulative scores [B, beam, T]).  Step 0 expands only the first beam (search.py:119-123); every later step takes the
    top `2 * beam` of beam x V candidates and keeps the first `beam` (none is ever finalised here)."""
    B, T, V = lprobs.shape
    beam = min(int(beam_size), V - 1)
    dev = lprobs.device
    tokens = torch.zeros(B, beam, T, dtype=torch.long, device=dev)
    scores = torch.zeros(B, beam, T, dtype=lprobs.dtype, device=dev)
    cum = torch.zeros(B, beam, dtype=lprobs.dtype, device=dev)
    for t in range(T):
        lp = lprobs[:, t]                                                # [B, V], the same for every beam
        if t == 0:
            cand = lp                                                    # only beam 0 is expanded at the first step
        else:
            cand = (cum[:, :, None] + lp[:, None, :]).reshape(B, beam * V)
        k = min(2 * beam, cand.size(1) - 1)
        top_s, top_i = torch.topk(cand, k=k, dim=1)
        sel_s, sel_i = top_s[:, :beam], top_i[:, :beam]
        src_beam, tok = sel_i // V, sel_i % V
        if t > 0:
            tokens[:, :, :t] = torch.gather(tokens[:, :, :t], 1, src_beam[:, :, None].expand(B, beam, t))
            scores[:, :, :t] = torch.gather(scores[:, :, :t], 1, src_beam[:, :, None].expand(B, beam, t))
        tokens[:, :, t] = tok
        scores[:, :, t] = sel_s
        cum = sel_s
    return tokens, scores
