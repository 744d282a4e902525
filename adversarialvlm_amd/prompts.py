"""Prompt / target pools.

The reference ships its pools as Python lists (`src/questions.py`, `src/answers.py`); their
text is research content that this repository deliberately does not copy.  The loop only
needs their SHAPE - a pool of user prompts, a pool of held-out test prompts, a pool of target
continuations - so the defaults here are neutral placeholders and real pools are loaded from
files the user supplies (`--questions_file`, `--test_questions_file`, `--answers_file`; one
entry per line, or a JSON list).
"""
import json
import os

DEFAULT_QUESTIONS = [f"Describe item number {i} in this picture." for i in range(1, 33)]
DEFAULT_TEST_QUESTIONS = [f"What is shown in region {i} of the image?" for i in range(1, 9)]
DEFAULT_ANSWERS = ["sure, here it is!"]


def load_pool(path, default):
    if not path:
        return list(default)
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    with open(path) as f:
        text = f.read()
    try:
        data = json.loads(text)
        if isinstance(data, list):
            return [str(x) for x in data]
    except json.JSONDecodeError:
        pass
    return [ln.strip() for ln in text.splitlines() if ln.strip()]
