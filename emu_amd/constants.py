"""Special tokens, placeholders and image statistics of Emu2.

Same names and values as the reference's ``Emu2/emu/constants.py:1-43`` so that code
written against ``emu.constants`` keeps working against ``emu_amd.constants``.
"""
EVA_IMAGE_SIZE = 448
OPENAI_DATASET_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_DATASET_STD = (0.26862954, 0.26130258, 0.27577711)

IGNORE_INDEX = -100

DEFAULT_PAD_TOKEN = "[PAD]"
DEFAULT_BOS_TOKEN = "<s>"
DEFAULT_EOS_TOKEN = "</s>"
DEFAULT_UNK_TOKEN = "<unk>"

DEFAULT_IMG_TOKEN = "[IMG]"
DEFAULT_IMG_END_TOKEN = "[/IMG]"
DEFAULT_IMAGE_TOKEN = "<image>"
DEFAULT_gIMG_TOKEN = "[gIMG]"
DEFAULT_gIMG_END_TOKEN = "[/gIMG]"
DEFAULT_EOC_TOKEN = "[EOC]"
DEFAULT_VIDEO_TOKEN = "[VIDEO]"

GRD_SYMBOL = "<grounding>"
BOP_SYMBOL = "<phrase>"
EOP_SYMBOL = "</phrase>"
BOO_SYMBOL = "<object>"
EOO_SYMBOL = "</object>"
DOM_SYMBOL = "</delimiter_of_multi_objects/>"
REC_SYMBOL = "<REC>"

USER_TOKEN = "[USER]"
ASSISTANT_TOKEN = "[ASSISTANT]"

DEFAULT_IMG_PLACEHOLDER = "[<IMG_PLH>]"
DEFAULT_VID_PLACEHOLDER = "[<VID_PLH>]"
FAKE_VIDEO_END_TOKEN = "[/VIDEO]"

GROUND_SYSTEM_MESSAGE = "You are a helpful assistant, dedicated to provide concise and efficient answers."
SYSTEM_MESSAGE = "You are a helpful assistant, dedicated to delivering comprehensive and meticulous responses."

# Token ids after the tokenizer extension (reference Emu2/emu/lm.py:43-63).
PAD_TOKEN_ID = 32000
BOS_TOKEN_ID = 1
EOS_TOKEN_ID = 2
IMG_TOKEN_ID = 32001        # [IMG]
IMG_END_TOKEN_ID = 32002    # [/IMG]
IMAGE_TOKEN_ID = 32003      # <image>
gIMG_TOKEN_ID = 32004       # [gIMG]
VOCAB_BASE = 32000
VOCAB_EMU2 = 32272          # Emu2 / Emu2-Gen
VOCAB_EMU2_CHAT = 32274     # Emu2-Chat (+[USER], [ASSISTANT])


def location_symbols(quantized_size=256, locate_special_token=2, flag_rec_symbol=True):
    """The 264 grounding symbols added after the 7 Emu tokens (reference lm.py:12-27)."""
    out = []
    if locate_special_token > 0:
        out.append(GRD_SYMBOL)
    out += [BOP_SYMBOL, EOP_SYMBOL, BOO_SYMBOL, EOO_SYMBOL, DOM_SYMBOL]
    if flag_rec_symbol:
        out.append(REC_SYMBOL)
    out += [f"<patch_index_{str(i).zfill(4)}>" for i in range(quantized_size + 1)]
    return out


def special_tokens_list(instruct: bool):
    """Order matters: ids are assigned consecutively from 32001 (reference lm.py:43-55)."""
    toks = [DEFAULT_IMG_TOKEN, DEFAULT_IMG_END_TOKEN, DEFAULT_IMAGE_TOKEN, DEFAULT_gIMG_TOKEN,
            DEFAULT_gIMG_END_TOKEN, DEFAULT_EOC_TOKEN, DEFAULT_VIDEO_TOKEN] + location_symbols()
    if instruct:
        toks += [USER_TOKEN, ASSISTANT_TOKEN]
    return toks
