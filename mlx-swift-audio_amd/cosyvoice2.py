"""Host-side mirror of CosyVoice2Model (TTS/CosyVoice2/CosyVoice2Model.swift:28-208) and of the tensor part of
prepareConditionals (TTS/CosyVoice2/CosyVoice2TTS.swift:370-430): generateTokens -> tokensToMel -> melToAudio, every stage on
the gfx950 HIP layer, including the CAM++ speaker embedding of the reference clip (speaker.py).  Text tokenisation stays with the
caller (CPU text code), so text arrives as token ids.
Every random draw of the reference is an explicit argument: `uniforms` (RAS sampler), `z` (CFM noise), `noise` (HiFT source)."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class CosyVoice2Conditionals:
    """CosyVoice2Conditionals (CosyVoice2Model.swift:557-575), tensors only."""
    prompt_speech_token: np.ndarray      # int32 [m]
    prompt_mel: np.ndarray               # float32 [2 m, 80] (time-major, as flow.inference consumes it)
    speaker_embedding: np.ndarray        # float32 [192]
    prompt_text: list[int]


class CosyVoice2Model:
    def __init__(self, ctx, llm, flow, hifigan, s3_tokenizer=None, speaker_encoder=None):
        self.ctx, self.llm, self.flow, self.hifigan, self.s3, self.speaker_encoder = ctx, llm, flow, hifigan, s3_tokenizer, speaker_encoder

    # ---- prepareConditionals, tensor part (CosyVoice2TTS.swift:383-423) -------------------------------------------------------------
    def prepare_conditionals(self, ref_wav: np.ndarray, speaker_embedding: np.ndarray | None = None, prompt_text=()):
        """ref_wav: 24 kHz mono reference clip (at most 30 s are used, CosyVoice2TTS.swift:375-380).  speaker_embedding None ->
        speakerEncoder(audio16k) (CosyVoice2TTS.swift:409): the CAM++ embedding of the 16 kHz clip."""
        from . import audio as A
        if self.s3 is None:
            raise ValueError("CosyVoice2Model was built without an S3 tokenizer")
        ref_wav_24k = np.ascontiguousarray(ref_wav, np.float32)[:30 * 24000]
        ref_wav_16k = A.resample_audio(self.ctx, ref_wav_24k, 24000, 16000)
        mel128 = A.s3_log_mel_spectrogram(self.ctx, ref_wav_16k, 128)
        tk, nt = self.s3.quantize(mel128[None], np.asarray([mel128.shape[1]], np.int32))
        toks = np.asarray(tk[0][:int(nt[0])], np.int32)
        mel80 = A.s3gen_mel_spectrogram(self.ctx, ref_wav_24k).T                       # [frames, 80]
        # the engine trims both so that mel frames == 2 * speech tokens (CosyVoice2TTS.swift:404-414)
        n = min(mel80.shape[0] // 2, toks.shape[0])
        if speaker_embedding is None:
            if self.speaker_encoder is None:
                raise ValueError("CosyVoice2Model was built without a speaker encoder and no speaker_embedding was given")
            speaker_embedding = self.speaker_encoder(ref_wav_16k)[0]
        return CosyVoice2Conditionals(toks[:n].copy(), np.ascontiguousarray(mel80[:2 * n]), np.ascontiguousarray(speaker_embedding, np.float32),
                                      list(prompt_text))

    # ---- the three stages (CosyVoice2Model.swift:53-133) ------------------------------------------------------------------------------
    def generate_tokens(self, text, prompt_text, prompt_speech_token, uniforms, sampling: int = 25, max_token_text_ratio: float = 20.0,
                        min_token_text_ratio: float = 2.0) -> list[int]:
        return self.llm.inference(list(text), list(prompt_text), list(prompt_speech_token), uniforms, max_token_text_ratio=max_token_text_ratio,
                                  min_token_text_ratio=min_token_text_ratio, top_k=sampling)

    def tokens_to_mel(self, tokens, prompt_token, prompt_feat, embedding, z, n_timesteps: int | None = None) -> np.ndarray:
        return self.flow.inference(tokens, prompt_token, prompt_feat, embedding, z, n_timesteps)

    def mel_to_audio(self, mel: np.ndarray, noise=None) -> np.ndarray:
        return self.hifigan(mel, noise=noise)[0]

    # ---- synthesize (CosyVoice2Model.swift:155-208) --------------------------------------------------------------------------------------
    def synthesize(self, text, cond: CosyVoice2Conditionals, uniforms, z_fn, noise_fn=None, sampling: int = 25, n_timesteps: int = 10,
                   max_token_text_ratio: float = 20.0, min_token_text_ratio: float = 2.0):
        """z_fn(T) -> [80, T] and noise_fn(L) -> [L, 9] supply the Gaussian draws once the generated length is known.
        Returns (audio [480 * 2 * n_tokens], tokens)."""
        tokens = self.generate_tokens(text, cond.prompt_text, cond.prompt_speech_token, uniforms, sampling, max_token_text_ratio, min_token_text_ratio)
        if not tokens:
            raise ValueError("No tokens generated")            # CosyVoice2Error.invalidInput (CosyVoice2Model.swift:182-184)
        T = 2 * (len(tokens) + len(cond.prompt_speech_token))
        mel = self.tokens_to_mel(np.asarray(tokens, np.int32), cond.prompt_speech_token, cond.prompt_mel, cond.speaker_embedding, z_fn(T), n_timesteps)
        noise = noise_fn(mel.shape[1] * self.hifigan.up) if noise_fn is not None else None
        return self.mel_to_audio(mel, noise), tokens

    def synthesize_batch(self, texts, cond: CosyVoice2Conditionals, uniforms, z_fn, noise_fn=None, sampling: int = 25, n_timesteps: int = 10,
                         max_token_text_ratio: float = 20.0, min_token_text_ratio: float = 2.0):
        """Several sentences of one speaker (the reference loops over them, CosyVoice2Model.swift:155-208): the LM runs sentence by sentence
        with its own uniform row, then ALL sentences go through one pass of the flow (mia_flow_inference_batch) and the vocoder turns each
        mel into audio.  Returns [(audio, tokens)] in order; every item equals synthesize() of that sentence with the same draws."""
        toks = [self.generate_tokens(t, cond.prompt_text, cond.prompt_speech_token, u, sampling, max_token_text_ratio, min_token_text_ratio)
                for t, u in zip(texts, uniforms)]
        if any(not t for t in toks):
            raise ValueError("No tokens generated")
        utts = [(np.asarray(t, np.int32), cond.prompt_speech_token, cond.prompt_mel, cond.speaker_embedding,
                 z_fn(2 * (len(t) + len(cond.prompt_speech_token)))) for t in toks]
        mels = self.flow.inference_batch(utts, n_timesteps)
        out = []
        for mel, t in zip(mels, toks):
            noise = noise_fn(mel.shape[1] * self.hifigan.up) if noise_fn is not None else None
            out.append((self.mel_to_audio(mel, noise), t))
        return out

    # ---- the other three modes (CosyVoice2Model.swift:253-397): the same three stages, different prompts ----------------------------------
    def _tokens_to_audio(self, tokens, cond, z_fn, noise_fn, n_timesteps):
        if not len(tokens):
            raise ValueError("No tokens generated")
        T = 2 * (len(tokens) + len(cond.prompt_speech_token))
        mel = self.tokens_to_mel(np.asarray(tokens, np.int32), cond.prompt_speech_token, cond.prompt_mel, cond.speaker_embedding, z_fn(T), n_timesteps)
        noise = noise_fn(mel.shape[1] * self.hifigan.up) if noise_fn is not None else None
        return self.mel_to_audio(mel, noise), list(tokens)

    def synthesize_cross_lingual(self, text, cond: CosyVoice2Conditionals, uniforms, z_fn, noise_fn=None, sampling: int = 25, n_timesteps: int = 10,
                                 max_token_text_ratio: float = 20.0, min_token_text_ratio: float = 2.0):
        """synthesizeCrossLingual (:253-310): the LM sees neither the prompt's text nor its speech tokens; the flow and the vocoder
        still condition on the reference clip."""
        tokens = self.generate_tokens(text, (), (), uniforms, sampling, max_token_text_ratio, min_token_text_ratio)
        return self._tokens_to_audio(tokens, cond, z_fn, noise_fn, n_timesteps)

    def synthesize_instruct(self, text, instruct_text, cond: CosyVoice2Conditionals, uniforms, z_fn, noise_fn=None, sampling: int = 25,
                            n_timesteps: int = 10, max_token_text_ratio: float = 20.0, min_token_text_ratio: float = 2.0):
        """synthesizeInstruct (:312-369): the instruction's ids take the prompt-text slot, no prompt speech tokens in the LM."""
        tokens = self.generate_tokens(text, instruct_text, (), uniforms, sampling, max_token_text_ratio, min_token_text_ratio)
        return self._tokens_to_audio(tokens, cond, z_fn, noise_fn, n_timesteps)

    def synthesize_vc(self, source_speech_token, cond: CosyVoice2Conditionals, z_fn, noise_fn=None, n_timesteps: int = 10):
        """synthesizeVC (:371-397): voice conversion -- the source clip's S3 tokens go straight to the flow with the target's conditionals."""
        return self._tokens_to_audio(list(np.asarray(source_speech_token).reshape(-1)), cond, z_fn, noise_fn, n_timesteps)[0]

    def tokenize_speech(self, wav_24k: np.ndarray) -> np.ndarray:
        """The source side of voice conversion (prepareSourceAudioForVC, CosyVoice2TTS.swift:624-653): 24 kHz clip -> S3 tokens."""
        from . import audio as A
        wav16 = A.resample_audio(self.ctx, np.ascontiguousarray(wav_24k, np.float32)[:30 * 24000], 24000, 16000)
        mel128 = A.s3_log_mel_spectrogram(self.ctx, wav16, 128)
        tk, nt = self.s3.quantize(mel128[None], np.asarray([mel128.shape[1]], np.int32))
        return np.asarray(tk[0][:int(nt[0])], np.int32)
