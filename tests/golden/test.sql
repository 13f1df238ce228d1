John	Smith	25	1
Jane	Doe	21	2
John	Smith	25	2
Jane	Doe	21	1
