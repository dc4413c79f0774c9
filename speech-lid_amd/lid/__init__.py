"""lid — drop-in mirror of the reference's Conformer LID pipeline (lid/ in kouyt5/speech-lid) on the lidk HIP engine."""
