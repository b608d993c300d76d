import re


def sent_tokenize(text):
    # Exact for the reference's hint template "The pose is X of a Y Z." joined by single spaces.
    return [s for s in re.split(r"(?<=[.!?])\s+", text.strip()) if s]
