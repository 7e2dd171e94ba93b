#!/usr/bin/env python3
"""Writes tests/golden/slt_corpus.json.

Source of the expectations: the reference's golden SQL tests
  tests/sqllogictest/indexing.slt:30-41        (8 9 4 1 7 2)
  tests/sqllogictest/bm25query.slt:35-82       (partial indexes: 8 4 2 / 9 1 7)
  tests/sqllogictest/fallback_parameter.slt:27-71 (limit 2 / 3 / 1)
for the query to_tsvector('english', 'PostgreSQL') over a 10-sentence corpus.

PostgreSQL's to_tsvector('english', ...) is not available in this environment, so
the token stream below is supplied BY HAND: default parser (hyphenated words emit
the compound and each part, each with its own position), english stop words
removed, Snowball-english-like stems.  Only two things influence the expected
orderings: which documents contain `postgresql` (tf = 1 in each) and every
document's length in lexeme positions (= sum of tf).  The exact spelling of the
other stems is immaterial.
"""
import json
import os

DOCS = {
    1: "postgresql power open-sourc open sourc object-rel object relat databas system 15 year "
       "activ develop",
    2: "full-text full text search techniqu search plain-text plain text document textual databas "
       "field postgresql support tsvector",
    3: "bm25 rank function use search engin estim relev document given search queri",
    4: "postgresql provid mani advanc featur like full-text full text search window function",
    5: "search rank databas import build effect inform retriev system",
    6: "bm25 rank algorithm deriv probabilist retriev framework",
    7: "full-text full text search index document allow fast text queri postgresql support gin "
       "gist index",
    8: "postgresql communiti activ regular improv databas system",
    9: "postgresql support non-rel non relat relat data type",
    10: "effect search rank algorithm bm25 improv search result understand relev",
}

EXPECT = [
    {"name": "indexing.slt:30-41 / bm25query.slt:35-45", "ids": "all", "k": 10,
     "order": [8, 9, 4, 1, 7, 2]},
    {"name": "bm25query.slt:54-63 partial index id%2=0", "ids": "even", "k": 10,
     "order": [8, 4, 2]},
    {"name": "bm25query.slt:72-82 partial index id%2=1", "ids": "odd", "k": 10,
     "order": [9, 1, 7]},
    {"name": "fallback_parameter.slt:27-35 limit=2", "ids": "all", "k": 2, "order": [8, 9]},
    {"name": "fallback_parameter.slt:37-47 bm25.limit=3", "ids": "all", "k": 3,
     "order": [8, 9, 4]},
    {"name": "fallback_parameter.slt:60-68 limit=1", "ids": "all", "k": 1, "order": [8]},
]

if __name__ == "__main__":
    out = {"query": ["postgresql"], "k1": 1.2, "b": 0.75,
           "docs": {str(i): s.split() for i, s in DOCS.items()}, "expect": EXPECT,
           "note": "token stream hand-supplied; see make_slt_fixture.py"}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "slt_corpus.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)
