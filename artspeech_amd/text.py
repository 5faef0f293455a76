"""Text front end of the inference script: the 178-symbol table and ``TextCleaner`` (test.py:19-38; the same table is
in meldataset.py:14-29 and Utils/RelTransformerEnc.py:6-10).  The vocabulary is data that a checkpoint's embedding
rows are indexed by, so it is reproduced exactly (pinned by tests/golden/text_golden.json, generated from the
reference); ``'`` occurs twice in the table and the later index wins, as in the reference's dict construction."""

_pad = "$"
_punctuation = ';:,.!?¡¿—…"«»“” '
_letters = 'ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz'
_letters_ipa = "ɑɐɒæɓʙβɔɕçɗɖðʤəɘɚɛɜɝɞɟʄɡɠɢʛɦɧħɥʜɨɪʝɭɬɫɮʟɱɯɰŋɳɲɴøɵɸθœɶʘɹɺɾɻʀʁɽʂʃʈʧʉʊʋⱱʌɣɤʍχʎʏʑʐʒʔʡʕʢǀǁǂǃˈˌːˑʼʴʰʱʲʷˠˤ˞↓↑→↗↘'̩'ᵻ"
symbols = [_pad] + list(_punctuation) + list(_letters) + list(_letters_ipa)
word_index_dictionary = {s: i for i, s in enumerate(symbols)}


class TextCleaner:
    """test.py:30-38: characters -> ids; unknown characters are printed and dropped (meldataset.py's variant raises)."""

    def __init__(self, dummy=None, strict=False):
        self.word_index_dictionary = word_index_dictionary
        self.strict = strict

    def __call__(self, text):
        indexes = []
        for char in text:
            try:
                indexes.append(self.word_index_dictionary[char])
            except KeyError:
                if self.strict:
                    raise
                print(char)
        return indexes
